"""inflatox_amd/_result_pool.py: result arrays are ordinary writeable arrays, a buffer is recycled only after the last
view of the result that used it is gone, and live results never share memory."""

import gc

import numpy as np

from inflatox_amd import _result_pool as pool


def _addr(a):
    return a.__array_interface__["data"][0]


def test_buffers_come_back_only_when_every_view_is_gone():
    gc.collect()
    a = pool.result_array((50, 40, 6))
    assert a.shape == (50, 40, 6) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    a[...] = 1.5
    views = (a[:, :, 0], a[:, :, 5])  # what complete_analysis returns: strided views of the (N0, N1, 6) array
    assert views[0].strides == (40 * 48, 48)
    addr = _addr(a)
    held = pool.held_bytes()
    del a
    gc.collect()
    assert pool.held_bytes() == held  # still referenced by the views
    b = pool.result_array((50, 40, 6))
    assert _addr(b) != addr and not np.shares_memory(b, views[0])
    assert float(views[1][3, 4]) == 1.5
    del views
    gc.collect()
    assert pool.held_bytes() == held + 50 * 40 * 48  # the first buffer is back
    c = pool.result_array((50, 40, 6))
    assert _addr(c) == addr  # and is reused for the next result of that size
    d = pool.result_array((40, 50, 6))  # same byte count, another shape: any free buffer of the size will do
    assert d.shape == (40, 50, 6) and not np.shares_memory(c, d) and not np.shares_memory(b, d)


def test_other_dtypes_sizes_and_the_limit():
    f = pool.result_array((30, 20), dtype=bool)
    assert f.dtype == np.bool_ and f.shape == (30, 20)
    f[...] = True
    assert f.all()
    assert pool.result_array((0, 7, 6)).shape == (0, 7, 6)
    big = pool.result_array((pool._LIMIT // 8 // 6 + 1, 1, 6))  # above half of the bound: a plain numpy array
    assert big.base is None or isinstance(big.base, np.ndarray) and big.base.base is None


def test_overflowing_the_pool_is_silent(recwarn):
    """More dropped results than the pool keeps per size: the surplus mappings are simply released."""
    arrays = [pool.result_array((64, 64, 6)) for _ in range(pool._PER_SIZE + 3)]
    held = pool.held_bytes()
    del arrays
    gc.collect()
    assert pool.held_bytes() <= held + pool._PER_SIZE * 64 * 64 * 48
    assert not [w for w in recwarn.list if "Unraisable" in str(w.category)]


def test_finalizer_never_takes_the_lock():
    """A result array can be finalised (cyclic GC) while this thread is inside the pool's locked region; the finalizer
    must not wait for the lock (a non-reentrant lock there hangs the process)."""
    a = pool.result_array((33, 31, 6))
    cycle = [a]
    cycle.append(cycle)  # only the cyclic collector can free it
    del a
    with pool._lock:
        del cycle
        gc.collect()  # runs _give_back on this thread, inside the locked region
    assert pool.held_bytes() >= 33 * 31 * 48


def test_stale_sizes_make_room_for_new_ones(monkeypatch):
    """Once the bound is reached the buffers that have been in the pool longest are released, so a scan that moves on
    to another grid size is pooled again."""
    gc.collect()
    pool.held_bytes()
    monkeypatch.setattr(pool, "_LIMIT", pool.held_bytes() + 3 * 100 * 100 * 48 + 1000)
    old = [pool.result_array((100, 100, 6)) for _ in range(3)]
    del old
    gc.collect()
    before = pool.held_bytes()
    new = pool.result_array((100, 101, 6))
    addr = _addr(new)
    del new
    gc.collect()
    assert pool.held_bytes() <= pool._LIMIT
    assert pool.held_bytes() >= before - 100 * 100 * 48  # at most one old buffer went
    again = pool.result_array((100, 101, 6))
    assert _addr(again) == addr  # the new size is pooled although the pool was full of the old one


def test_default_bound_follows_the_machine_and_release_empties_the_pool(monkeypatch):
    """Large results are where recycling pays most (dropping a 768 MiB result costs more than the sweep that made it): the default
    bound scales with the memory of the machine / cgroup, between 1 and 16 GiB; INFLX_RESULT_POOL_MB overrides; release() returns
    everything."""
    monkeypatch.delenv("INFLX_RESULT_POOL_MB", raising=False)
    assert (1 << 30) <= pool._default_limit() <= (16 << 30)
    monkeypatch.setenv("INFLX_RESULT_POOL_MB", "3")
    assert pool._default_limit() == 3 << 20
    monkeypatch.setenv("INFLX_RESULT_POOL_MB", "0")
    assert pool._default_limit() == 0
    a = pool.result_array((77, 13, 6))
    addr = _addr(a)
    del a
    gc.collect()
    assert pool.held_bytes() >= 77 * 13 * 48
    assert pool.release() >= 77 * 13 * 48 and pool.held_bytes() == 0
    b = pool.result_array((77, 13, 6))  # a fresh mapping, usable as ever
    b[...] = 2.0
    assert float(b.sum()) == 2.0 * 77 * 13 * 6 and (addr or True)


def test_memory_limits_are_found_on_every_level_up_to_the_root(tmp_path):
    """A memory limit on an intermediate ancestor (a systemd slice, SLURM's job / step levels) bounds the pool too: the walk goes from
    the process's own cgroup up to the root and keeps every limit; "max" and the v1 "unlimited" sentinel are none."""
    from inflatox_amd import _result_pool as rp

    assert list(rp._levels("/a/b/c")) == ["/a/b/c", "/a/b", "/a", ""] and list(rp._levels("/")) == [""]
    v2, v1 = tmp_path / "v2", tmp_path / "v1"
    (v2 / "job_7" / "step_0" / "task_3").mkdir(parents=True)
    (v2 / "memory.max").write_text("max\n")
    (v2 / "job_7" / "memory.max").write_text(str(6 << 30) + "\n")  # the job's limit: two levels above the process
    (v2 / "job_7" / "step_0" / "memory.max").write_text("max\n")
    (v2 / "job_7" / "step_0" / "task_3" / "memory.max").write_text(str(64 << 30) + "\n")
    (v1 / "slurm" / "uid_1" / "job_9").mkdir(parents=True)
    (v1 / "memory.limit_in_bytes").write_text(str(rp._V1_UNLIMITED) + "\n")
    (v1 / "slurm" / "memory.limit_in_bytes").write_text(str(48 << 30) + "\n")
    (v1 / "slurm" / "uid_1" / "job_9" / "memory.limit_in_bytes").write_text(str(rp._V1_UNLIMITED) + "\n")
    proc = tmp_path / "cgroup"
    proc.write_text("0::/job_7/step_0/task_3\n5:cpu,memory:/slurm/uid_1/job_9\n3:cpuset:/\n")
    limits = rp.cgroup_memory_limits(str(proc), str(v2), str(v1))
    assert sorted(limits) == [6 << 30, 48 << 30, 64 << 30] and min(limits) == 6 << 30
    # inside a container the process's path does not exist below the mount: the root file still counts
    proc.write_text("0::/docker/abcdef\n")
    (v2 / "memory.max").write_text(str(32 << 30) + "\n")
    assert rp.cgroup_memory_limits(str(proc), str(v2), str(v1)) == [32 << 30]
    # no /proc/self/cgroup at all: the roots
    assert rp.cgroup_memory_limits(str(tmp_path / "nothing"), str(v2), str(v1)) == [32 << 30]
