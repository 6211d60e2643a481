"""N>1 path on CPU: world_size-2 gloo run of the shipped sharding / 4-step exchange logic (toyni_amd/dist.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_sharding_and_fourstep(world):
    import __graft_entry__ as entry
    entry.build_hip()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert f"DIST OK {world}" in res.stdout


def test_shard_batch_properties():
    from toyni_amd.dist import shard_batch
    for total in (0, 1, 7, 1024, 1027):
        for world in (1, 2, 3, 8):
            shards = [shard_batch(total, world, r) for r in range(world)]
            assert shards[0][0] == 0 and sum(c for _, c in shards) == total
            for (s0, c0), (s1, _) in zip(shards, shards[1:]):
                assert s1 == s0 + c0
            assert max(c for _, c in shards) - min(c for _, c in shards) <= 1
