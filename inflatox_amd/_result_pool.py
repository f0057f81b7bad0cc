"""Host memory of the result arrays the front-end returns, recycled between calls.

The reference allocates ``np.zeros((N0, N1, 6))`` per call (python/inflatox/consistency_conditions.py:290); every page
of such an array is faulted in while it is filled and unmapped again when the caller drops the result.  For the
reference's default 1000 x 1000 grid (48 MB) that page work is three quarters of a call here (4.5 ms against 1.0 ms into
memory that is already resident, scripts/latency_probe.py): the sweep itself takes 30 microseconds.  A parameter scan --
call, look at the result, drop it, call again -- therefore gets its arrays from a small pool: when the last view of a
result dies, its (page-resident) buffer goes back to the pool instead of to the kernel, and the next result of the same
size is built on it.  The arrays are ordinary writeable numpy arrays; nothing is shared between live results; the sweep
writes every element, so no zeroing is needed.

Large results gain the most: dropping a 768 MiB result (4096 x 4096) costs 37-47 ms on a GPU box's host -- the kernel tears the
mapping down through the GPU driver's MMU notifiers, the pages having been the target of a DMA -- against 17 ms for the call that
produced it, and filling fresh pages costs another 2.5 ms (scripts/big_result_probe.py).

``INFLX_RESULT_POOL_MB`` bounds the memory the pool keeps (default: an eighth of the machine's -- or the cgroup's -- memory, at
least 1 GiB and at most 16 GiB: an 8192 x 8192 result of 3.2 GB is recycled on any host a GPU sits in); 0 switches the pool off
(plain ``np.zeros``).  Arrays above half of the bound are never pooled.  ``release()`` hands everything the pool holds back to the
system.  When a returning buffer does not fit, the buffers that have been lying
in the pool longest are released first (a scan that moves on to another grid size does not stay stuck with the old one).

Locking: the finalizer of a result array can run at any point at which the garbage collector runs -- also while this
module holds its lock on the same thread (an allocation inside the locked region may start a collection that finalises
another pooled array sitting in a reference cycle).  The finalizer therefore takes no lock at all: it appends the buffer
to a ``collections.deque`` (atomic under the GIL), and ``result_array`` / ``held_bytes`` move the returned buffers into
the size-indexed free lists under the (re-entrant) lock.
"""

from __future__ import annotations

import collections
import mmap
import os
import threading
import weakref

import numpy as np



_V1_UNLIMITED = 0x7FFFFFFFFFFFF000  # what cgroup v1 reports for "no limit" (PAGE_COUNTER_MAX rounded to pages)


def _levels(path: str):
    """A cgroup path and every ancestor of it: "/a/b/c" -> "/a/b/c", "/a/b", "/a", ""."""
    path = path.rstrip("/")
    while True:
        yield path
        if not path:
            return
        path = path.rsplit("/", 1)[0]


def cgroup_memory_limits(proc_cgroup: str = "/proc/self/cgroup", v2_mount: str = "/sys/fs/cgroup", v1_mount: str = "/sys/fs/cgroup/memory") -> list[int]:
    """Every memory limit in force on the way from this process's cgroup up to the hierarchy root (bytes; "max" and the v1
    "unlimited" sentinel are no limits).  Inside a container the process's path usually does not exist below the mount: the levels
    that are missing are skipped and the root file still counts."""
    paths = {"v2": [""], "v1": [""]}
    try:
        with open(proc_cgroup) as fh:
            for line in fh:
                _, controllers, path = line.rstrip("\n").split(":", 2)
                if controllers == "":
                    paths["v2"] = list(_levels(path))
                elif "memory" in controllers.split(","):
                    paths["v1"] = list(_levels(path))
    except (OSError, ValueError):
        pass
    limits = []
    for files in ([f"{v2_mount}{lvl}/memory.max" for lvl in paths["v2"]], [f"{v1_mount}{lvl}/memory.limit_in_bytes" for lvl in paths["v1"]]):
        for name in files:
            try:
                with open(name) as fh:
                    value = int(fh.read().strip())
            except (OSError, ValueError):  # no such level here, or "max"
                continue
            if 0 < value < _V1_UNLIMITED:
                limits.append(value)
    return limits


def _default_limit() -> int:
    env = os.environ.get("INFLX_RESULT_POOL_MB")
    if env is not None:
        try:
            return max(0, int(env.strip())) << 20
        except ValueError:  # a malformed value must not break `import inflatox_amd`: fall through to the default
            pass
    try:
        memory = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        memory = 8 << 30
    # a container's (or a batch job's) share of it: cgroup v2 `memory.max`, cgroup v1 `memory.limit_in_bytes` (what SLURM's
    # task/cgroup plugin and older container runtimes set).  A limit may sit on ANY level between the process's own cgroup and the
    # root -- a systemd slice, SLURM's job and step levels --: walk upwards and take the smallest
    memory = min([memory] + cgroup_memory_limits())
    return max(1 << 30, min(memory // 8, 16 << 30))


_LIMIT = _default_limit()
_PER_SIZE = 4

_lock = threading.RLock()
_returned: collections.deque = collections.deque()  # (buf, nbytes) pushed by finalizers, no lock taken
_free: dict[int, collections.deque] = {}  # nbytes -> deque of (age, buf), oldest first
_held = 0
_age = 0


def _give_back(buf, nbytes: int) -> None:
    """Called when the last numpy view of a pooled buffer has died.  Lock-free on purpose (see the module text)."""
    _returned.append((buf, nbytes))


def _evict_oldest(keep_size: int) -> bool:
    """Release the buffer that has been in the pool longest (any size but ``keep_size`` first).  Lock held."""
    global _held
    best = None
    for size, lst in _free.items():
        if lst and size != keep_size and (best is None or lst[0][0] < _free[best][0][0]):
            best = size
    if best is None and _free.get(keep_size):
        best = keep_size
    if best is None:
        return False
    _free[best].popleft()  # the mapping is unmapped when this last reference goes
    _held -= best
    return True


def _drain() -> None:
    """Move the buffers finalizers have returned into the free lists.  Lock held."""
    global _held, _age
    while True:
        try:
            buf, nbytes = _returned.popleft()
        except IndexError:
            return
        lst = _free.get(nbytes)
        if lst is None:
            lst = _free[nbytes] = collections.deque()
        if len(lst) >= _PER_SIZE or nbytes > _LIMIT // 2:
            continue  # not pooled: unmapped when `buf` goes out of scope (no explicit close(): the dying owner array may still hold its buffer export)
        while _held + nbytes > _LIMIT and _evict_oldest(nbytes):
            pass
        if _held + nbytes <= _LIMIT:
            _age += 1
            lst.append((_age, buf))
            _held += nbytes


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
        _drain()
        lst = _free.get(nbytes)
        if lst:
            buf = lst.pop()[1]  # the most recently returned one: its pages are the likeliest to be resident
            _held -= nbytes
    if buf is None:
        buf = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    owner = np.frombuffer(buf, dtype=dtype, count=count)  # every view's .base collapses to this array
    weakref.finalize(owner, _give_back, buf, nbytes)
    return owner.reshape(shape)


def release() -> int:
    """Hand every buffer the pool holds back to the system; returns the number of bytes released.  (Live results are not
    touched; their buffers come back to the pool when they die, as always.)"""
    global _held
    with _lock:
        _drain()
        freed = _held
        _free.clear()
        _held = 0
    return freed


def held_bytes() -> int:
    with _lock:
        _drain()
        return _held
