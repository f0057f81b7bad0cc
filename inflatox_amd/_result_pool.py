"""Host memory of the result arrays the front-end returns, recycled between calls.

The reference allocates ``np.zeros((N0, N1, 6))`` per call (python/inflatox/consistency_conditions.py:290); every page
of such an array is faulted in while it is filled and unmapped again when the caller drops the result.  For the
reference's default 1000 x 1000 grid (48 MB) that page work is three quarters of a call here (4.5 ms against 1.0 ms into
memory that is already resident, scripts/latency_probe.py): the sweep itself takes 30 microseconds.  A parameter scan --
call, look at the result, drop it, call again -- therefore gets its arrays from a small pool: when the last view of a
result dies, its (page-resident) buffer goes back to the pool instead of to the kernel, and the next result of the same
size is built on it.  The arrays are ordinary writeable numpy arrays; nothing is shared between live results; the sweep
writes every element, so no zeroing is needed.

``INFLX_RESULT_POOL_MB`` (default 1024) bounds the memory the pool keeps; 0 switches it off (plain ``np.zeros``).
Arrays above half of the bound are never pooled.
"""

from __future__ import annotations

import mmap
import os
import threading
import weakref

import numpy as np

_LIMIT = max(0, int(os.environ.get("INFLX_RESULT_POOL_MB", "1024"))) << 20
_PER_SIZE = 4

_lock = threading.Lock()
_free: dict[int, list] = {}
_held = 0


def _give_back(buf, nbytes: int) -> None:
    """Called when the last numpy view of a pooled buffer has died."""
    global _held
    with _lock:
        lst = _free.setdefault(nbytes, [])
        if len(lst) < _PER_SIZE and _held + nbytes <= _LIMIT:
            lst.append(buf)
            _held += nbytes
    # not pooled: the mapping is unmapped when this last reference goes (no explicit close(): the dying owner array may
    # still hold its buffer export at this point)


def result_array(shape, dtype=np.float64) -> np.ndarray:
    """A writeable C-contiguous array of this shape whose memory comes from the pool when one of the right size is free
    (contents arbitrary: the caller fills every element), else from a fresh anonymous mapping."""
    global _held
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    nbytes = count * dtype.itemsize
    if _LIMIT == 0 or nbytes == 0 or nbytes > _LIMIT // 2:
        return np.zeros(shape, dtype=dtype)
    buf = None
    with _lock:
        lst = _free.get(nbytes)
        if lst:
            buf = lst.pop()
            _held -= nbytes
    if buf is None:
        buf = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    owner = np.frombuffer(buf, dtype=dtype, count=count)  # every view's .base collapses to this array
    weakref.finalize(owner, _give_back, buf, nbytes)
    return owner.reshape(shape)


def held_bytes() -> int:
    with _lock:
        return _held
