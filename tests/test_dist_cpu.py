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
