"""The library's batch-sized buffers inside torch's caching allocator.

Workspaces, training tapes and backward scratch are gigabytes at training batch sizes.  As raw ``hipMalloc``'s they compete with
PyTorch's caching allocator for the same HBM from the outside: under a real training run (ResNet / CLIP encoders around the
denoiser) the allocator holds most of the device as cached-but-free blocks, HIP reports almost nothing free, and the library's
allocation fails although the memory is idle -- ``torch.cuda.empty_cache()`` only helps while no cached block is partly in use.
``install()`` hands the library ``torch.cuda.caching_allocator_alloc`` / ``caching_allocator_delete`` through the plain-C hook
``mdt_set_allocator`` (include/mdt_hip.h): the buffers then live in torch's pool, show up in ``torch.cuda.memory_allocated()``
and obey torch's own out-of-memory handling.  The library frees a buffer only after ``hipDeviceSynchronize()`` (growth) or in
its destroy functions, so the stream a block was allocated on never matters.  ``MDT_TORCH_ALLOCATOR=0`` keeps raw hipMalloc.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from .. import _lib

_ALLOC_T = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
_FREE_T = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_lock = threading.Lock()
_installed = None  # keeps the ctypes callbacks alive for the life of the process


def install() -> bool:
    """Idempotent; returns whether the torch allocator is in place."""
    global _installed
    if os.environ.get("MDT_TORCH_ALLOCATOR", "1") in ("0", ""):
        return False
    with _lock:
        if _installed is not None:
            return True
        import torch

        def _alloc(nbytes, _user):
            try:
                return torch.cuda.caching_allocator_alloc(int(nbytes))  # current device, current stream
            except Exception:  # torch.cuda.OutOfMemoryError and friends: the library reports "out of memory"
                return None

        def _free(ptr, _user):
            try:
                torch.cuda.caching_allocator_delete(ptr)
            except Exception:
                pass

        a, f = _ALLOC_T(_alloc), _FREE_T(_free)
        lib = _lib.load()
        _lib.check(lib.mdt_set_allocator(a, f, None))
        _installed = (a, f)
        # The callbacks must outlive every handle: a module global alone can be cleared at interpreter shutdown before a
        # surviving HipEngine.__del__ / resampler destroy runs, and the library would call a freed ffi closure.  Pin them on
        # the CDLL object AND leak one reference each (never collected); at exit switch the library back to its own
        # hipMalloc / hipFree pair -- buffers that came from torch are then dropped without a call into a half torn-down torch
        # (mdt_allocator_detach, include/mdt_hip.h).
        lib._mdt_allocator_callbacks = (a, f)
        C.pythonapi.Py_IncRef(C.py_object(a))
        C.pythonapi.Py_IncRef(C.py_object(f))
        import atexit
        atexit.register(_uninstall_at_exit, lib)
        return True


def _uninstall_at_exit(lib) -> None:
    try:
        lib.mdt_allocator_detach()
    except Exception:  # pragma: no cover -- shutdown path
        pass
