#!/usr/bin/env python3
"""Benchmark of the TriangleMix hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic input:
the packed block-sparse index is built on the device and the masked
(triangle) attention of one Qwen3-8B layer runs over the whole prompt
(prefill, paged K/V cache, page size 16, shuffled block table).

N = 1: one 32768-token prompt, 32 query heads / 8 K/V heads / head dim 128.
N > 1: one rank per GPU, head-parallel tensor parallelism -- rank r owns H/N
query heads and Hkv/N K/V heads -- with the global batch grown to N prompts,
so each rank keeps the same number of (prompt, head) units (weak scaling).
The ranks come either from torch.distributed.run (RANK / WORLD_SIZE in the
environment) or, when `--gpus N` is given without them, from this script
itself: it starts N copies of itself BEFORE touching any GPU, relays rank 0's
line and exits non-zero unless all N ranks finish.  The attention needs no
collective; the output projection's all-reduce (the only exchange step of the
layer) is timed on every run with more than one rank (and with --oproj on one)
and reported beside the main value (extra.oproj_gemm_allreduce_ms,
extra.layer_ms_with_oproj), never inside it.

--model picks the head shape: 8b (32/8 heads, 32k), 32b (64/8 heads, 32k: 16/2
per GPU at --gpus 4), 72b-fp8 (64/8 heads, fp8 e4m3 K/V cache, 64k: 8/1 per GPU
at --gpus 8).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (no sparsity)
QWEN3_8B = dict(num_heads=32, num_kv_heads=8, head_dim=128, hidden=4096)
# head shapes of the benchmark configurations (the 72B row is the 64-head / 8-K/V-head shape with hidden 8192;
# there is no such Qwen3 release, the shape is what matters to the attention)
MODELS = {
    "8b": dict(name="Qwen3-8B", num_heads=32, num_kv_heads=8, head_dim=128, hidden=4096, seq_len=32768, kv="bf16"),
    "32b": dict(name="Qwen3-32B", num_heads=64, num_kv_heads=8, head_dim=128, hidden=5120, seq_len=32768, kv="bf16"),
    "72b-fp8": dict(name="Qwen3-72B-shape", num_heads=64, num_kv_heads=8, head_dim=128, hidden=8192, seq_len=65536,
                    kv="fp8"),
}
PATTERN = (4, 32, 64)  # sink, window, last: the reference defaults
PAGE = 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed steps run for this long before the warm-up steps: an idle GPU needs tens of "
                         "milliseconds of load before its clocks settle (0.25 ms steps: 10 warm-up steps are 2.5 ms)")
    ap.add_argument("--model", choices=sorted(MODELS), default="8b", help="head shape and K/V cache type")
    ap.add_argument("--seq-len", type=int, default=0, help="prompt length (0: the model's configuration)")
    ap.add_argument("--index-in-launch", action="store_true",
                    help="the attention call builds the step's index itself, inside its launch where it can "
                         "(default: a build launch ahead of the attention)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo, size their shard of the job on the host and print "
                         "the line without a measurement (checks the launcher and the sharding arithmetic)")
    ap.add_argument("--oproj", action="store_true", help="also time o_proj GEMM + all-reduce (reported separately)")
    ap.add_argument("--no-baselines", action="store_true", help="skip the dense and CPU baselines")
    ap.add_argument("--cpu-rows", type=int, default=4096, help="query rows of the CPU-baseline sample")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearsal on one GPU: rank 0's share of an N-GPU job (N prompts, heads / N), no process group")
    ap.add_argument("--chunk-tiles", type=int, default=0,
                    help="piece length of the dense rows in key tiles (0: the library's advice for the batch)")
    ap.add_argument("--cpu-heads", type=int, default=8, help="heads of the CPU-baseline sample")
    ap.add_argument("--kv-cache", choices=("model", "bf16", "fp8"), default="model",
                    help="K/V cache type (default: the model configuration's); an A/B aid, named in config.workload")
    ap.add_argument("--no-band", action="store_true",
                    help="A/B aid: every q-block through the work list (no band workgroups for the sparse q-blocks)")
    return ap.parse_args()


def make_inputs(ops, dev, n, batch, heads, kv_heads, dim, seed, fp8=False):
    """q [batch*n, heads, dim] bf16, paged K/V cache (bf16, or fp8 e4m3 with unit scales) filled through the
    cache-write kernel."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    T = batch * n
    q = torch.randn((T, heads, dim), generator=g, dtype=torch.float32).to(dev).bfloat16()
    k = torch.randn((T, kv_heads, dim), generator=g, dtype=torch.float32).to(dev).bfloat16()
    v = torch.randn((T, kv_heads, dim), generator=g, dtype=torch.float32).to(dev).bfloat16()
    pages_per_seq = (n + PAGE - 1) // PAGE
    nblk = batch * pages_per_seq
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    block_table = perm.view(batch, pages_per_seq).contiguous()
    pos = torch.arange(n, dtype=torch.int64)
    slots = torch.cat([block_table[b].to(torch.int64)[pos // PAGE] * PAGE + pos % PAGE for b in range(batch)])
    if fp8:
        kv_cache = torch.zeros((2, nblk, PAGE, kv_heads, dim), dtype=torch.uint8, device=dev)
        ops.reshape_and_cache_fp8(k, v, kv_cache[0], kv_cache[1], slots.to(dev), 1.0, 1.0)
    else:
        kv_cache = torch.zeros((2, nblk, PAGE, kv_heads, dim), dtype=torch.bfloat16, device=dev)
        ops.reshape_and_cache(k, v, kv_cache[0], kv_cache[1], slots.to(dev))
    cu_h = (np.arange(batch + 1) * n).astype(np.int32)
    sl_h = np.full((batch,), n, np.int32)
    return dict(q=q, k=k, v=v, kv_cache=kv_cache, block_table=block_table.to(dev), cu_h=cu_h, sl_h=sl_h,
                cu=torch.from_numpy(cu_h).to(dev), sl=torch.from_numpy(sl_h).to(dev))


def allowed_pairs(n, si, sl, last, triangle):
    """Number of (query, key) pairs the pattern keeps for one n-token prompt (closed form of the row counts)."""
    i = np.arange(n, dtype=np.int64)
    causal = i + 1
    if not triangle:
        return int(causal.sum())
    sparse = np.minimum(causal, (si + 1) + (sl + 1))  # sink columns + window; they overlap only while i - sl <= si + 1
    return int(np.where(n - i < last, causal, sparse).sum())


def cpu_baseline(n, heads, dim, rows, sample_heads):
    """Times the C restatement of the reference's CPU path (mask loops + dense
    eager attention) on a bounded sample: `rows` query rows spread evenly over
    the n-token prompt, all n keys, `sample_heads` heads; scaled to all rows
    and heads.  Baseline only."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libtmx_oracle.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    rng = np.random.default_rng(0)
    step = max(1, n // rows)
    rows = min(rows, n)
    row0 = n - 1 - (rows - 1) * step  # includes the last row
    k = rng.standard_normal((n, sample_heads, dim), dtype=np.float32)
    v = rng.standard_normal((n, sample_heads, dim), dtype=np.float32)
    q = rng.standard_normal((rows, sample_heads, dim), dtype=np.float32)
    mask = np.empty((rows, n), np.float32)
    out = np.empty_like(q)
    t0 = time.perf_counter()
    lib.tmx_oracle_triangle_additive_rows(mask.ctypes.data_as(fp), n, *PATTERN, row0, step, rows)
    t_mask = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc = lib.tmx_oracle_eager_attention_rows(q.ctypes.data_as(fp), k.ctypes.data_as(fp), v.ctypes.data_as(fp),
                                             mask.ctypes.data_as(fp), out.ctypes.data_as(fp), rows, n, sample_heads,
                                             dim, ctypes.c_float(dim ** -0.5))
    t_attn = time.perf_counter() - t0
    assert rc == 0 and np.isfinite(out).all()
    # whole job: the mask is built once (all n rows), the attention runs for all rows and heads
    t_full = t_mask * (n / rows) + t_attn * (n / rows) * (heads / sample_heads)
    return {"value": n / t_full, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{rows} of {n} query rows (every {step}th, incl. the last) x all {n} keys x {sample_heads} of "
                      f"{heads} heads, fp32, mask loops {t_mask:.2f}s + dense attention {t_attn:.2f}s, scaled to "
                      f"all rows and heads"}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpu_count():
    """GPUs a child process will see, found WITHOUT any HIP / HSA call (the launcher must stay GPU-free: it goes on
    to start one process per GPU).  The visibility variables win when set; otherwise the kernel driver's topology
    is read (/sys/class/kfd: one node per agent, GPUs are the nodes with SIMDs).  None: unknown (no such files) --
    the ranks then find out themselves and a rank without a device fails the job."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        if var in os.environ:
            counts.append(len([x for x in os.environ[var].split(",") if x.strip()]))
    if counts:
        return min(counts)
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        found = 0
        for d in os.listdir(nodes):
            try:
                with open(os.path.join(nodes, d, "properties")) as f:
                    props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                found += 1
        return found
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """`--gpus N` without a launcher: start N copies of this script, one rank per GPU, BEFORE anything here
    has touched a GPU (this process never does: it only relays rank 0's line).  Returns the exit code:
    non-zero unless all N ranks ran to the end and rank 0 printed a line for N GPUs.  Every rank's stderr goes
    to a file of its own; when a rank fails, the end of ITS stderr is shown."""
    import tempfile
    n = args.gpus
    seen = None if args.dry_run else visible_gpu_count()
    if seen is not None and seen < n:
        print(f"bench.py: --gpus {n} but only {seen} HIP device(s) are visible", file=sys.stderr)
        return 2
    port = free_port()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(tempfile.TemporaryFile(mode="w+", prefix=f"bench_rank{r}_"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r], text=True))

    def stderr_tail(r, lines=25):
        errs[r].seek(0)
        return "".join(errs[r].readlines()[-lines:])

    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:  # the exact processes started above, nothing else
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        print(f"bench.py: rank {failed} of {n} exited with {procs[failed].returncode}; no result.  Its stderr ends:\n"
              f"{stderr_tail(failed)}", file=sys.stderr)
        others = [r for r, p in enumerate(procs) if r != failed and p.returncode not in (0, -9)]
        if others:
            print(f"bench.py: ranks {others} failed as well (rank {others[0]}'s stderr ends:\n{stderr_tail(others[0], 8)})",
                  file=sys.stderr)
        return 1
    out = procs[0].stdout.read()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if not lines or json.loads(lines[-1]).get("n_gpus") != n:
        print(f"bench.py: rank 0 printed no line for {n} GPUs", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def dry_run(args, cfg, n, world, rank):
    """No GPU: rendezvous over gloo, every rank sizes its shard of the job on the host (head shard, index sizes,
    piece length), the ranks agree on the maximum, rank 0 prints the line without a measurement."""
    import torch.distributed as dist
    from vllm_trianglemix_amd import ops
    from vllm_trianglemix_amd.parallel import shard_heads
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = shard_heads(cfg["num_heads"], cfg["num_kv_heads"], rank, world)
    H, Hkv = shard.num_heads, shard.num_kv_heads
    batch = world
    cu_h, sl_h = (np.arange(batch + 1) * n).astype(np.int32), np.full((batch,), n, np.int32)
    bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
    chunk = ops.suggest_chunk_tiles(cu_h, sl_h, *PATTERN, bm, bn, ops.MODES["triangle"], ops.units_per_item(H, Hkv),
                                    num_cus=256)
    sz = ops.index_sizes(cu_h, sl_h, *PATTERN, bm, bn, ops.MODES["triangle"], chunk_tiles=chunk)
    t = torch.tensor([float(sz.num_items * ops.units_per_item(H, Hkv)), float(H), float(Hkv)], dtype=torch.float64)
    extra = {}
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        # the layer's exchange step on host tensors over gloo, at a toy size: the row-parallel output projection with
        # its chunked all-reduce (the path the measured run takes on the GPUs over RCCL), checked against the sum of
        # every rank's product formed locally
        from vllm_trianglemix_amd.parallel import RowParallelOutputProjection
        D, hidden, toks = cfg["head_dim"], 64, 200
        def part(r):
            g = torch.Generator().manual_seed(100 + r)
            return torch.randn((toks, H, D), generator=g), torch.randn((hidden, H * D), generator=g) * 0.05
        x, w = part(rank)
        proj = RowParallelOutputProjection(w, chunk_tokens=64)
        t0 = time.perf_counter()
        y = proj(x)
        dt = time.perf_counter() - t0
        want = sum(xr.reshape(toks, -1) @ wr.t() for xr, wr in (part(r) for r in range(world)))
        ok = torch.tensor([1.0 if torch.allclose(y, want, rtol=1e-4, atol=1e-4) else 0.0], dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() != 1.0:
            raise SystemExit("bench.py --dry-run: the chunked o_proj all-reduce does not equal the sum of the ranks' products")
        extra = {"oproj_gemm_allreduce_ms": dt * 1e3, "layer_ms_with_oproj": None,
                 "oproj": {"tokens": toks, "hidden": hidden, "chunk_tokens": 64, "backend": "gloo (dry run, toy size)",
                           "allreduce_bytes": toks * hidden * 4}}
    if rank == 0:
        print(json.dumps({"metric": metric_name(cfg, n), "value": None, "unit": "tokens/s", "n_gpus": world, "steps": 0,
                          "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "dry_run": True, "extra": extra,
                          "config": {"workload": workload_name(cfg, n), "global_batch": batch, "seq_len": n,
                                     "heads_per_gpu": int(t[1]), "kv_heads_per_gpu": int(t[2]),
                                     "head_dim": cfg["head_dim"], "parallelism": f"tp{world} (heads)",
                                     "work_units_per_gpu": int(t[0]), "chunk_tiles": chunk}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def metric_name(cfg, n):
    return (f"prefill tokens/sec (TriangleMix attention op, index build included), {cfg['name']} "
            f"{n // 1024 if n % 1024 == 0 else n}{'k' if n % 1024 == 0 else '-token'} ctx")


def workload_name(cfg, n):
    return (f"{cfg['name']} {'fp8 K/V cache, bf16 arithmetic' if cfg['kv'] == 'fp8' else 'bf16'} seq_len={n} TriangleMix "
            f"prefill (one attention layer: block-index build + masked attention), sink/window/last={PATTERN}, "
            f"paged KV page {PAGE}")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    cfg = dict(MODELS[args.model])
    if args.kv_cache != "model":
        cfg["kv"] = args.kv_cache
    n = args.seq_len or cfg["seq_len"]
    if args.dry_run:
        return dry_run(args, cfg, n, world, rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the TriangleMix path has no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from vllm_trianglemix_amd import ops
    from vllm_trianglemix_amd._lib import (TMX_ATTN_DECODE_ONLY, TMX_ATTN_NO_BAND, TMX_ATTN_ONLY_COMBINE,
                                           TMX_ATTN_SHORT_QUERIES, TMX_ATTN_SKIP_COMBINE)
    base_flags = TMX_ATTN_NO_BAND if args.no_band else 0
    from vllm_trianglemix_amd.parallel import RowParallelOutputProjection, shard_heads

    fp8 = cfg["kv"] == "fp8"
    job_world = args.emulate_world if (args.emulate_world > 0 and world == 1) else world
    shard = shard_heads(cfg["num_heads"], cfg["num_kv_heads"], rank, job_world)
    H, Hkv, D = shard.num_heads, shard.num_kv_heads, cfg["head_dim"]
    batch = job_world  # weak scaling: one more prompt per GPU
    inp = make_inputs(ops, dev, n, batch, H, Hkv, D, seed=0, fp8=fp8)  # same prompts on every rank, different head shards
    si, sl, last = PATTERN
    bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
    scale = D ** -0.5
    out = torch.empty_like(inp["q"])
    kc, vc = inp["kv_cache"][0], inp["kv_cache"][1]

    def build(mode, pages=True, defer=False):
        # pages of every listed tile are resolved by the builder (block table lookups leave the attention loop);
        # the piece length is the library's advice for this batch (one piece of the dense rows per CU: 64 tiles here)
        if mode == "triangle" and args.chunk_tiles > 0:
            chunk = args.chunk_tiles
        else:
            chunk = ops.suggest_chunk_tiles(inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn, ops.MODES[mode],
                                            ops.units_per_item(H, Hkv))
        if not pages:
            return ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn,
                                         ops.MODES[mode], chunk_tiles=chunk, defer=defer)
        return ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn,
                                     ops.MODES[mode], chunk_tiles=chunk, block_table=inp["block_table"],
                                     page_size=PAGE, defer=defer)

    idx0 = build("triangle")
    work = ops.alloc_workspace(idx0, H, D, dev)

    # (an fp8 K/V cache is read in place: the kernel descales inside its tile loop)
    def attend(idx, flags=0):
        ops.attention(inp["q"], kc, vc, out, idx, inp["cu"], inp["sl"], scale, block_table=inp["block_table"],
                      workspace=work, flags=flags | base_flags)

    # One step = the index of the step built from the batch description + the attention over it: a build launch,
    # then the attention.  --index-in-launch: the attention call builds the index itself (TMX_ATTN_BUILD_INDEX;
    # workgroup 0 of the launch builds it while the band workgroups, which do not read it, already stream) -- the
    # same index and output; measured no faster at 32k tokens and slower at 8k (the work-list workgroups, which
    # have to wait for it, are on the launch's critical path), so it is not the default.
    def step():
        attend(build("triangle", defer=args.index_in_launch))

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    # an idle GPU first: its clocks settle after tens of milliseconds of load, far more than a few 0.25 ms steps
    # (measured on one box: 248 us per step after 10 warm-up steps, 207 us after 200)
    t_settle = time.perf_counter() + args.settle_ms * 1e-3
    while time.perf_counter() < t_settle:
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    tokens_per_step = batch * n  # whole job: every prompt is processed once per step (heads are split over ranks)
    value = tokens_per_step / (elapsed / args.steps)

    extra = {}
    # ---- dominant kernel alone, timed with events on the launch stream (ahead of the projection below: a large
    # GEMM leaves the clocks elsewhere than a run of attention steps does) ----
    def timed(fn, iters, median=False):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(dev)
        ts = [a.elapsed_time(b) for a, b in evs]
        return float(np.median(ts) if median else np.mean(ts))  # ms

    def timed_back_to_back(fn, iters, group=10):
        """Average duration of one launch of `fn` out of `group` launches that run back to back.

        A pair of events around ONE launch of a 10 us kernel measures the events and the host's launch path as much
        as the kernel.  Here the stream is first held by a spinning kernel while the host queues an event, `group`
        launches and a second event behind it; once the spin ends they run with nothing but the dispatch gap between
        them, and the events' distance / group is the launch's duration as a kernel trace reports it.
        """
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        # spin length for about 1.5 ms (the host queues `group` calls in a fraction of that), from a timed spin
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.cuda._sleep(1_000_000)
        b.record()
        torch.cuda.synchronize(dev)
        spin = int(1_000_000 * 1.5 / max(a.elapsed_time(b), 1e-3))
        for a, b in evs:
            torch.cuda._sleep(spin)
            a.record()
            for _ in range(group):
                fn()
            b.record()
            torch.cuda.synchronize(dev)
        return float(np.mean([a.elapsed_time(b) for a, b in evs])) / group  # ms

    def attn_only():
        attend(idx0, TMX_ATTN_SKIP_COMBINE)

    def combine_only():
        attend(idx0, TMX_ATTN_ONLY_COMBINE)

    iters = max(5, min(args.steps, 50))
    for _ in range(3):
        attn_only()
    reps = max(3, iters // 5)
    t_attn_ms = timed_back_to_back(attn_only, reps)
    t_comb_ms = timed_back_to_back(combine_only, reps)
    t_index_ms = timed_back_to_back(lambda: build("triangle"), reps)
    # algorithmic bytes per token per attention call: Q read + O write (16-bit) + K read + V read (16-bit, or one
    # byte each out of an fp8 cache)
    bytes_per_token = 2 * H * D * 2 + 2 * Hkv * D * (1 if fp8 else 2)
    algo_bytes = bytes_per_token * batch * n
    achieved = algo_bytes / (t_attn_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of the same kernel on the same workload, from two rocprofv3 --pmc
    # passes (FETCH_SIZE x 2 + WRITE_SIZE; tools/profile_round.sh writes the file); null when absent
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    # (the counters were taken on the default workload: one 32k prompt, Qwen3-8B heads, 16-bit cache)
    if os.path.exists(tpath) and world == 1 and job_world == 1 and args.model == "8b" and batch == 1 and not fp8:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("seq_len") == n and tj.get("kernel") == "attn_kernel":
            traffic = tj.get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "attn_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)" if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": t_attn_ms, "combine_ms": t_comb_ms, "index_build_ms": t_index_ms,
                "index_build": "by the attention call (inside its launch where the step allows)" if args.index_in_launch
                               else "a launch of its own"}

    # ---- the layer's only exchange step: row-parallel o_proj GEMM + all-reduce of its partial products over the
    # ranks (linear.py:1358-1359; RCCL over xGMI), chunked over tokens so that a chunk's all-reduce runs beside the
    # next chunk's GEMM.  Always timed when there is more than one rank (every rank takes part), reported beside
    # the attention-only `value`, never inside it.
    if world > 1 or args.oproj:
        hidden = cfg["hidden"]
        w = (torch.randn((hidden, H * D), device=dev) * 0.02).bfloat16()
        chunk_tokens = 8192
        proj = RowParallelOutputProjection(w, chunk_tokens=chunk_tokens if world > 1 else 0)
        for _ in range(2):
            y = proj(out)
        sync_all()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            y = proj(out)
        sync_all()
        t_op = (time.perf_counter() - t0) / reps
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([t_op], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_op = float(tt.item())
        extra["oproj_gemm_allreduce_ms"] = t_op * 1e3
        # (serial sum: the attention of a layer is followed by its projection; no overlap between the two is assumed)
        extra["layer_ms_with_oproj"] = ms_per_step + t_op * 1e3
        extra["oproj"] = {"tokens": int(out.shape[0]), "hidden": hidden, "chunk_tokens": chunk_tokens if world > 1 else 0,
                          "backend": "nccl (RCCL)" if world > 1 else "none (one rank: GEMM only)",
                          "allreduce_bytes": int(out.shape[0]) * hidden * 2 if world > 1 else 0,
                          "gemm_TFLOPs": 2.0 * out.shape[0] * hidden * H * D / t_op / 1e12 if world == 1 else None}
        del y, w, proj

    if not args.no_baselines and rank == 0 and not fp8:
        # dense-causal baselines at the same shape (attention-op level)
        try:
            # (the index a causal layer's step gets from the backend: the dense causal kernel computes the prompt
            # without an index, so the index leaves it out -- no tile list, no partial-result slots)
            idx_c = ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], 0, 0, 0, bm, bn,
                                          ops.MODES["causal"], chunk_tiles=64, block_table=inp["block_table"],
                                          page_size=PAGE, leave_dense=True)
            # (with the staging area the library sizes for long prompts: K/V rows gathered once per call, V as fp16,
            # and the kernel's tiles then come into LDS by LDS-DMA; the gather is inside the timed call)
            work_c = ops.alloc_workspace(idx_c, H, D, dev, num_kv_heads=Hkv)

            def dense_own():
                ops.attention(inp["q"], kc, vc, out, idx_c, inp["cu"], inp["sl"], scale,
                              block_table=inp["block_table"], workspace=work_c)

            # (a 9 ms launch heats the chip within a few calls: median of 7 after 2 untimed ones)
            dense_own()
            dense_own()
            t_dense = timed(dense_own, 7, median=True)
            extra["dense_causal_own_kernel_tokens_per_s"] = batch * n / (t_dense * 1e-3)
            # matrix-core side: algorithmic flops = 4 * D * heads * kept (query, key) pairs (QK^T and PV, 2 flops each)
            fl_dense = 4.0 * D * H * batch * allowed_pairs(n, si, sl, last, False)
            fl_tri = 4.0 * D * H * batch * allowed_pairs(n, si, sl, last, True)
            extra["mfma"] = {"peak_TFLOPs": MFMA_PEAK_TFLOPS,
                             "dense_causal_TFLOPs": fl_dense / (t_dense * 1e-3) / 1e12,
                             "dense_causal_frac_of_peak": fl_dense / (t_dense * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                             "triangle_TFLOPs": fl_tri / (t_attn_ms * 1e-3) / 1e12,
                             "triangle_frac_of_peak": fl_tri / (t_attn_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS}
            extra["mfma"]["dense_causal_form"] = "staged (LDS-DMA tiles)" if len(work_c) > 2 else "register"
            extra["speedup_vs_own_dense"] = t_dense / (ms_per_step)
            # attention of the whole model with the reference's preset for this size (trianglemix_config.py:65-72:
            # 8 triangle layers for an 8B model, 12 for 32B; the other layers attend causally above 2048 tokens)
            layers, tri_layers = {"8b": (36, 8), "32b": (64, 12)}.get(args.model, (80, 8))
            t_model = tri_layers * ms_per_step + (layers - tri_layers) * t_dense
            extra["whole_model_attention"] = {"layers": layers, "triangle_layers": tri_layers, "ms": t_model,
                                              "tokens_per_s": batch * n / (t_model * 1e-3),
                                              "all_dense_ms": layers * t_dense}
            del work_c
        except Exception as e:  # pragma: no cover
            extra["dense_causal_own_kernel_error"] = repr(e)
        try:
            import torch.nn.functional as F
            qs = inp["q"][:n].transpose(0, 1).unsqueeze(0)
            ks = inp["k"][:n].transpose(0, 1).unsqueeze(0)
            vs = inp["v"][:n].transpose(0, 1).unsqueeze(0)

            def sdpa():
                return F.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True)

            sdpa()
            t_sdpa = timed(sdpa, 3)
            extra["dense_causal_torch_sdpa_tokens_per_s"] = n / (t_sdpa * 1e-3)
            extra["speedup_vs_torch_sdpa"] = (t_sdpa * batch) / ms_per_step
        except Exception as e:  # pragma: no cover
            extra["dense_causal_torch_sdpa_error"] = repr(e)[:200]

    if not args.no_baselines and rank == 0 and world == 1 and not fp8:
        # decode step of the same model (every decode row is a dense last row of the pattern): 64 requests with
        # 8192 cached tokens each, one query row per request; the kernel streams the K/V of every request once
        try:
            Bd, Nd = 64, 8192
            gd = torch.Generator().manual_seed(1)
            nblk = Bd * (Nd // PAGE)
            kcd = torch.randn((nblk, PAGE, Hkv, D), generator=gd).to(dev).bfloat16()
            vcd = torch.randn((nblk, PAGE, Hkv, D), generator=gd).to(dev).bfloat16()
            btd = torch.randperm(nblk, generator=gd).to(torch.int32).view(Bd, -1).to(dev)
            qd = torch.randn((Bd, H, D), generator=gd).to(dev).bfloat16()
            outd = torch.empty_like(qd)
            cu_dh, sl_dh = np.arange(Bd + 1, dtype=np.int32), np.full((Bd,), Nd, np.int32)
            cu_d, sl_d = torch.from_numpy(cu_dh).to(dev), torch.from_numpy(sl_dh).to(dev)
            chunk_d = ops.suggest_chunk_tiles(cu_dh, sl_dh, si, sl, last, bm, bn, ops.MODES["triangle"],
                                              ops.units_per_item(H, Hkv))
            idx_d = ops.build_block_index(cu_d, sl_d, cu_dh, sl_dh, si, sl, last, bm, bn, ops.MODES["triangle"],
                                          chunk_tiles=chunk_d, block_table=btd, page_size=PAGE)
            work_d = ops.alloc_workspace(idx_d, H, D, dev)
            # the flags the backend passes for a step of one-row requests: the decode kernel runs alone
            dec_flags = TMX_ATTN_SHORT_QUERIES | TMX_ATTN_DECODE_ONLY

            def decode():
                ops.attention(qd, kcd, vcd, outd, idx_d, cu_d, sl_d, scale, block_table=btd, workspace=work_d,
                              flags=dec_flags)

            for _ in range(5):
                decode()
            t_dec = timed(decode, 20)
            kv_bytes = 2 * Bd * Nd * Hkv * D * 2
            extra["decode_64x8192_ms"] = t_dec
            extra["decode_64x8192_kv_GBps"] = kv_bytes / (t_dec * 1e-3) / 1e9
            extra["decode_64x8192_frac_of_hbm_peak"] = kv_bytes / (t_dec * 1e-3) / 1e9 / HBM_PEAK_GBS
            # the same step out of an fp8 (e4m3) cache: half the bytes, descaled inside the tile loop
            k8, v8 = torch.zeros(kcd.shape, dtype=torch.uint8, device=dev), torch.zeros(kcd.shape, dtype=torch.uint8, device=dev)
            ops.reshape_and_cache_fp8(kcd.view(-1, Hkv, D), vcd.view(-1, Hkv, D), k8, v8,
                                      torch.arange(kcd.shape[0] * PAGE, device=dev), 1.0, 1.0)

            def decode8():
                ops.attention(qd, k8, v8, outd, idx_d, cu_d, sl_d, scale, block_table=btd, workspace=work_d,
                              flags=dec_flags)

            for _ in range(5):
                decode8()
            t_dec8 = timed(decode8, 20)
            extra["decode_64x8192_fp8_cache_ms"] = t_dec8
            extra["decode_64x8192_fp8_cache_kv_GBps"] = 0.5 * kv_bytes / (t_dec8 * 1e-3) / 1e9
            del kcd, vcd, work_d, k8, v8
        except Exception as e:  # pragma: no cover
            extra["decode_error"] = repr(e)[:200]

    if not args.no_baselines and rank == 0 and world == 1:
        # The headline is a settled-clock figure.  A serving step after a pause sees an idle GPU's clocks: one
        # step, timed by events, after 200 ms without work (median of 5).
        cold = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            time.sleep(0.2)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            step()
            eb.record()
            torch.cuda.synchronize(dev)
            cold.append(ea.elapsed_time(eb))
        extra["step_after_200ms_idle_ms"] = float(np.median(cold))

    cpu = None
    if not args.no_baselines and rank == 0 and world == 1:
        cpu = cpu_baseline(n, cfg["num_heads"], D, args.cpu_rows, args.cpu_heads)

    if rank == 0:
        line = {
            "metric": metric_name(cfg, n),
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "settle_ms": args.settle_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload_name(cfg, n),
                       "global_batch": batch, "seq_len": n, "heads_per_gpu": H, "kv_heads_per_gpu": Hkv,
                       "head_dim": D, "parallelism": f"tp{world} (heads)" if world > 1 else
                       (f"rank 0 of an emulated tp{job_world} job" if job_world > 1 else "single GPU")},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
