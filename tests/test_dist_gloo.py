"""N > 1 path on CPU: a world_size-2 gloo job running bench.py's own control flow (bench.run) with
the oracle standing in for the GPU decoder, plus unit checks of the partition / reduction helpers
(espflix_amd/dist.py).  The data path has no collective; the tests check that the shards tile the
stream id range (weak and fixed-batch partitions), that the gathered per-stream chain hashes equal
the golden table of the reference decoder, and that a corrupted stream aborts the run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from espflix_amd import dist as edist

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


def test_fixed_batch_partition_is_floor_k_R_over_total():
    for total in (8192, 10, 7):
        for world in (1, 2, 3, 4, 8):
            owner = np.arange(total) * world // total          # stream k -> rank floor(k R / total)
            for r in range(world):
                lo, hi = edist.shard_fixed(r, world, total)
                assert list(np.nonzero(owner == r)[0]) == list(range(lo, hi))


def test_checksum_is_order_independent_and_sensitive():
    h = np.random.default_rng(1).integers(0, 2**63, 100).astype(np.uint64)
    c = edist.frame_checksum(h)
    assert c == edist.frame_checksum(h[::-1])
    h2 = h.copy()
    h2[17] ^= np.uint64(1)
    assert c != edist.frame_checksum(h2)


def _run_job(world, extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), "--gpus", str(world)] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


def test_two_rank_gloo_job_runs_the_bench_control_flow():
    p = _run_job(2, ["--streams", "3", "--steps", "2", "--warmup", "1", "--fixed-batch", "9", "--cpu-baseline-seconds", "0.2"], 29533)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "gloo saw 2 ranks" in p.stderr
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["streams_total"] == 6
    assert out["parity_gate"]["passed"] and out["parity_gate"]["streams_checked"] == 6
    assert out["fixed_batch_8192"]["streams_total"] == 9 and out["fixed_batch_8192"]["scaling"] == "strong"
    assert out["fixed_batch_8192"]["streams_per_gpu"] in (4, 5)      # ceil / floor split of 9 over 2 ranks
    assert out["cpu_baseline"] is not None and out["cpu_baseline"]["value"] > 0   # rank 0, N > 1 too
    # a call that runs as two groups of streams is 24 k_recon launches per step: per-launch figures follow
    r = out["roofline"]
    assert r["launches_per_step"] == 24 and abs(r["avg_launch_ms"] * 24 - r["stage_ms"]["k_recon x12"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) < 1e-6
    # value = pictures of ALL ranks / max-over-ranks time
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - 6 * 12) < 1e-6
    # the same job in one process: same per-stream hashes, hence the same checksum of checksums
    q = _run_job(1, ["--streams", "6", "--steps", "1", "--warmup", "0", "--no-fixed-batch", "--no-other-workloads", "--no-cpu-baseline"], 29534)
    assert q.returncode == 0, q.stderr[-3000:]
    one = json.loads([l for l in q.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert one["checksum_of_checksums"] == out["checksum_of_checksums"]


def test_bench_aborts_when_a_stream_differs_from_the_reference(tmp_path):
    # a decoder stand-in that flips one bit of one frame hash on rank 1 must not produce a number
    worker = tmp_path / "bad_worker.py"
    worker.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np, dist_worker\n"
        "orig = dist_worker.OracleDecoder.frame_hashes\n"
        "def bad(self):\n"
        "    h = orig(self)\n"
        "    if os.environ.get('RANK') == '1': h[1, 0] ^= np.uint64(1)\n"
        "    return h\n"
        "dist_worker.OracleDecoder.frame_hashes = bad\n"
        "dist_worker.main()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29535", str(worker), "--gpus", "2", "--streams", "2", "--steps", "1", "--warmup", "0",
           "--no-fixed-batch", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "parity gate" in p.stderr and "RESULT" not in p.stdout
