"""The one tolerance rule of the attention parity tests, and a record of what every check measured.

The bar (fp32 kernel output against the fp64 oracle on the same 16-bit-rounded inputs):

    |got - ref| <= 1e-3 * max(1, |ref|)  [+ 2^-8 * |ref| when the kernel's output is bf16 / fp16]

1e-3 absolute is the bar the path is built to; above |ref| = 1 it turns relative, because a sparse row averages a
handful of values and the probabilities of the fp16 product carry 11 bits (tests plant V entries of 40000 on
purpose).  A 16-bit output adds half an output ulp of its own rounding (2^-8 relative covers bf16 and fp16).

Every call of :func:`assert_close` records the largest error it saw, as a multiple of the bar and in absolute
terms, under its ``what`` label (the kernel or route the rows went through); ``conftest.py`` prints the table at the
end of the session and writes it to ``gpurun_out/parity_errors.json``.
"""
import numpy as np

ATOL = 1e-3
OUT16_REL = 2.0 ** -8

MEASURED = {}  # what -> dict(max_abs=..., max_frac_of_bar=..., checks=...)


def bar(ref, out16=False):
    """Per-element tolerance for reference values ``ref``."""
    a = np.abs(np.nan_to_num(np.asarray(ref, dtype=np.float64)))
    t = ATOL * np.maximum(1.0, a)
    if out16:
        t = t + a * OUT16_REL
    return t


def record(what, max_abs, max_frac):
    m = MEASURED.setdefault(what, dict(max_abs=0.0, max_frac_of_bar=0.0, checks=0))
    m["max_abs"] = max(m["max_abs"], float(max_abs))
    m["max_frac_of_bar"] = max(m["max_frac_of_bar"], float(max_frac))
    m["checks"] += 1


def assert_close(got, ref, what, out16=False, info=None, nan_equal=False):
    """``got`` within the bar of ``ref`` everywhere; ``what`` names the kernel / route for the error record.
    ``nan_equal``: positions that are NaN in both (rows the pattern leaves without any key) count as equal."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if out16:
        what += " [16-bit output]"
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if nan_equal:
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (what, info)
        both = np.isnan(ref)
        err = np.abs(np.where(both, 0.0, got - np.where(both, 0.0, ref)))
    else:
        assert not np.isnan(got).any(), (what, info, "NaN in the kernel's output")
        err = np.abs(got - ref)
    if err.size == 0:
        return 0.0
    frac = err / bar(ref, out16)
    at = np.unravel_index(int(frac.argmax()), frac.shape)
    record(what, err.max(), frac.max())
    print(f"[parity] {what}: max |err| {err.max():.2e}, {frac.max():.2f} of the bar")
    assert frac.max() <= 1.0, (what, info, f"max err {err[at]:.3e} at {at}: ref {ref[at]:.6g} got {got[at]:.6g}")
    return float(err.max())
