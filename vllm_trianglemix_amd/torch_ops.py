"""``torch.ops._trianglemix_C.*``: the TORCH_LIBRARY registration of the C ABI
(``csrc/torch_bindings.cpp``), the form in which the reference loads its own
native ops (vllm/platforms/rocm.py:32-40 imports the extension module;
vllm/_custom_ops.py:153-196 wraps ``torch.ops._rocm_C.*``).  The ctypes
binding in ``_lib.py`` and this one call the same library entry points."""
from __future__ import annotations

import importlib
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
BINDING_PATH = os.path.join(_HERE, "_trianglemix_C.so")
_loaded = False


def load():
    """Registers the ops (once) and returns ``torch.ops._trianglemix_C``."""
    global _loaded
    if not _loaded:
        if not os.path.exists(BINDING_PATH):
            raise ImportError(f"{BINDING_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}`")
        # a plain import: the shared object is a Python extension module whose load registers the ops
        # (PyInit__trianglemix_C, the reference's REGISTER_EXTENSION, csrc/core/registration.h:13-27)
        importlib.import_module(__package__ + "._trianglemix_C")
        _loaded = True
    return torch.ops._trianglemix_C
