"""N > 1 path on CPU: world_size-2 gloo job running the partition + reduction code bench.py uses
(espflix_amd/dist.py).  The data path has no collective; the test checks that the shards tile
the stream id range and that the gathered checksum equals the single-process checksum."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from espflix_amd import dist as edist
from espflix_amd import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_tile_the_id_range():
    for world in (1, 2, 4, 8):
        ids = []
        for r in range(world):
            first, n = edist.shard(r, world, 1024)
            ids.extend(range(first, first + n))
        assert ids == list(range(world * 1024))
    with pytest.raises(ValueError):
        edist.shard(2, 2, 4)


def test_checksum_is_order_independent_and_sensitive():
    h = np.random.default_rng(1).integers(0, 2**63, 100).astype(np.uint64)
    c = edist.frame_checksum(h)
    assert c == edist.frame_checksum(h[::-1])
    h2 = h.copy()
    h2[17] ^= np.uint64(1)
    assert c != edist.frame_checksum(h2)


def test_two_rank_gloo_job_matches_single_process():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py"), "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) == 2.0            # max over ranks of (1 + rank)
    assert int(line[3]) == 2 * 3 * 4
    b = gen.Batch(0, 6, 4, 12, 0, 1)        # the same six streams in one process
    hashes = np.concatenate([oracle.decode(b.es(i), 0)[1] for i in range(6)])
    assert line[2] == f"{edist.frame_checksum(hashes):016x}"
