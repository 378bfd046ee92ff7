"""N>1 path on the CPU: world_size-2 (and 3, ragged) gloo rehearsal of the protocol bench.py runs over RCCL —
rank-0 weight load, one broadcast, contiguous batch sharding with no data-path collective, max-over-ranks timing."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range(pkg):
    from mbn_amd_pkg import dist as mdist
    for total, world in ((2048, 8), (256, 1), (10, 3), (2, 4), (0, 2)):
        spans = [mdist.shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert mdist.shard_range(2048, 8, 3) == (768, 1024)       # BASELINE config 4: 8 x 256
    with pytest.raises(ValueError):
        mdist.shard_range(8, 2, 2)


def test_c_shard_range_matches_python(pkg):
    """The C host's multi-GPU mode (mobilenet --gpus G, include/mbn.h mbn_shard_range) cuts the batch exactly like the
    torch.distributed form (dist.py shard_range): contiguous, total/world each, the first total%world ranks one more."""
    import ctypes as C
    from mbn_amd_pkg import dist as mdist
    lib = pkg.host_lib()
    for total, world in ((2048, 8), (256, 1), (10, 3), (2, 4), (0, 2), (257, 8), (7, 7)):
        for r in range(world):
            f, c = C.c_int(), C.c_int()
            assert lib.mbn_shard_range(total, world, r, C.byref(f), C.byref(c)) == 0
            lo, hi = mdist.shard_range(total, world, r)
            assert (f.value, f.value + c.value) == (lo, hi)
    f, c = C.c_int(), C.c_int()
    for bad in ((8, 2, 2), (8, 0, 0), (8, 2, -1), (-1, 2, 0)):
        assert lib.mbn_shard_range(*bad, C.byref(f), C.byref(c)) == pkg.EINVAL
    assert lib.mbn_shard_range(8, 2, 0, None, C.byref(c)) == pkg.EINVAL


@pytest.mark.parametrize("world,total", [(2, 6), (3, 7)])
def test_gloo_broadcast_and_sharded_forward(pkg, orc, tmp_path, world, total):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path), str(total)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(world)]
    assert res[0]["match"] is True                                      # concat of shards == full-batch forward
    assert len({x["blob_sum"] for x in res}) == 1                       # everybody holds rank 0's parameters
    assert [x["lo"] for x in res] == sorted(x["lo"] for x in res) and res[-1]["hi"] == total
    slow = {round(x["slowest"], 9) for x in res}
    assert len(slow) == 1 and abs(res[0]["slowest"] - max(x["dt"] for x in res)) < 1e-9   # MAX over ranks


def test_rank_thread_skeleton_with_fake_jobs():
    """mbn_run_ranks / mbn_rank_barrier (host/mbn_ranks.c), the thread skeleton of `mobilenet --gpus G`, with three fake jobs on
    the CPU: (1) every rank runs and the barrier really separates the phases; (2) a thread that cannot be
    created (simulated at rank 1 and at rank 2) makes the call fail with NO job run and without hanging (bare
    pthread_create + pthread_barrier stranded the started ranks); (3) a rank that fails before the barrier releases its peers
    with an error instead of leaving them blocked."""
    import ctypes as C
    import threading
    import time
    from mbn_amd import import_package
    pkg = import_package()
    lib = pkg.host_lib()
    events, lock = [], threading.Lock()

    def job(rank, arg, sync):
        time.sleep(0.02 * (2 - rank))                      # rank 0 arrives last
        with lock:
            events.append(("a", rank))
        if lib.mbn_rank_barrier(sync) != 0:
            return pkg.EDEVICE
        with lock:
            events.append(("b", rank))
        return 0
    fn = pkg.RANK_FN(job)
    rcs = (C.c_int * 3)()
    assert lib.mbn_run_ranks(3, fn, None, -1, rcs) == 0 and list(rcs) == [0, 0, 0]
    assert sorted(events[:3]) == [("a", 0), ("a", 1), ("a", 2)] and sorted(events[3:]) == [("b", 0), ("b", 1), ("b", 2)]
    for fail_at in (1, 2):
        events.clear()
        t0 = time.time()
        assert lib.mbn_run_ranks(3, fn, None, fail_at, rcs) == pkg.ENOMEM
        assert events == [] and list(rcs) == [pkg.EUNSUPPORTED] * 3 and time.time() - t0 < 5.0

    def failing(rank, arg, sync):
        if rank == 1:
            return pkg.EIO                                   # leaves before the barrier
        return lib.mbn_rank_barrier(sync)                    # must come back with an error, not block
    t0 = time.time()
    assert lib.mbn_run_ranks(3, pkg.RANK_FN(failing), None, -1, rcs) != 0
    assert rcs[1] == pkg.EIO and rcs[0] == pkg.EDEVICE and rcs[2] == pkg.EDEVICE and time.time() - t0 < 5.0
    assert lib.mbn_run_ranks(0, fn, None, -1, None) == pkg.EINVAL


def test_bench_refuses_more_ranks_than_gpus_before_touching_a_device():
    """`python bench.py --gpus N` (no launcher in front) starts its own ranks as a child process; with fewer than N visible GPUs
    and no rehearsal override it must say so and leave with ENODEV instead of spawning ranks that die in hipSetDevice. Here
    (no GPU at all) that is the whole observable behaviour; the launch itself is a GPU test."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 19 and "MBN_ENODEVICE" in r.stderr and r.stdout.strip() == ""
