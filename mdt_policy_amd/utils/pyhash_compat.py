"""``pyhash.fnv1_32`` as the reference's harness uses it, over the library's host helper ``mdt_fnv1_32``.

Reference: the vendored pyhash 0.9.3 (``pyhash-0.9.3/src``): ``fnv1_32_t`` (FNV1.h:18-39) is a ``Hasher``
(Hash.h:102-176) whose ``__call__(*args, seed=...)`` folds every argument into a running value that STARTS AT THE
SEED (default 0, Hash.h:113,123,167) -- not at the FNV offset basis -- through ``fnv_32_buf``
(fnv/hash_32.c:91-113).  A Python 3 ``str`` argument is hashed as its UTF-16 code units in native byte order
without a BOM (Hash.h:241-268), ``bytes`` as they are.

Call sites mirrored: ``hasher = pyhash.fnv1_32()``; ``hasher(str(idx)) % window_range``
(mdt/datasets/base_dataset.py:20-37) and ``hasher(str(initial_condition.values()))`` (mdt/evaluation/utils.py:305).
"""
from __future__ import annotations

import sys

from .. import _lib

_UTF16 = "utf-16-le" if sys.byteorder == "little" else "utf-16-be"


class fnv1_32:  # noqa: N801 - the reference's (pyhash's) spelling
    def __init__(self, seed: int = 0):
        self.seed = int(seed) & 0xFFFFFFFF

    def __call__(self, *args, seed=None) -> int:
        lib = _lib.load()
        value = self.seed if seed is None else int(seed) & 0xFFFFFFFF
        for a in args:
            if isinstance(a, str):
                data = a.encode(_UTF16, "surrogatepass")
            elif isinstance(a, (bytes, bytearray, memoryview)):
                data = bytes(a)
            else:
                raise TypeError(f"unsupported argument type {type(a).__name__}")
            value = int(lib.mdt_fnv1_32(data, len(data), value))
        return value


hasher = fnv1_32()


def get_validation_window_size(idx: int, min_window_size: int, max_window_size: int) -> int:
    """Deterministic window size of a validation sequence (reference mdt/datasets/base_dataset.py:24-37)."""
    window_range = max_window_size - min_window_size + 1
    return min_window_size + hasher(str(idx)) % window_range
