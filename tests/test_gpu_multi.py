"""The multi-device entry points of the C-ABI (efx_multi_*: one context and one host thread per device, streams dealt in
contiguous blocks) and the C++ scaling harness built on them (tools/efx_scale.cpp, RCCL all-gather of the
chain hashes).  The development lease has one GPU: two contexts on the same device exercise the partition, and the
harness runs with one device."""
import os
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def efx():
    import espflix_amd
    espflix_amd.load_library()
    return espflix_amd


@pytest.mark.parametrize("devices,n", [([0], 5), ([0, 0], 7), ([0, 0, 0], 8), ([0, 0], 1)])
def test_batch_dealt_over_contexts_equals_oracle(efx, devices, n):
    from espflix_amd import gen
    b = gen.Batch(50, n, 12, 12, 0)
    streams = b.all_es()
    m = efx.MultiDecoder(devices, max_streams=(n + len(devices) - 1) // len(devices), max_pictures=12, ring_depth=2,
                         max_stream_bytes=sum(s.size for s in streams) + 4096)
    m.upload(streams)
    for call in range(2):   # the ring rotates the same way on every device
        m.decode()
    counts, status = m.results()
    assert (counts == 12).all() and not status.any()
    hashes = m.frame_hashes()
    firsts = [efx.partition_first(n, len(devices), r) for r in range(len(devices) + 1)]
    for k in range(n):
        dev, local = m.locate(k)
        assert firsts[dev] <= k < firsts[dev + 1] and local == k - firsts[dev]
        _, oh, _, _ = oracle.decode(streams[k], 0)
        # two calls of 12 pictures on the reference's double buffer: pictures 10 and 11 are what is left
        assert sorted(int(x) for x in hashes[k]) == sorted(int(x) for x in oh[10:12])
    m.close()


def test_capacity_is_per_device(efx):
    from espflix_amd import gen
    streams = gen.Batch(0, 5, 2, 12, 0).all_es()
    m = efx.MultiDecoder([0, 0], max_streams=2, max_pictures=2)
    with pytest.raises(efx.EfxError) as e:
        m.upload(streams)   # 3 + 2
    assert e.value.status == -4
    m.upload(streams[:4])
    m.decode()
    assert (m.results()[0] == 2).all()
    m.close()


def test_cpp_scaling_harness_gathers_through_rccl_and_matches_the_reference():
    exe = os.path.join(ROOT, "tools", "efx_scale")
    if not os.path.exists(exe):
        pytest.skip("tools/efx_scale not built (make scale)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([exe, "--devices", "1", "--streams", "128", "--steps", "5", "--golden",
                        os.path.join(ROOT, "tests", "golden", "bench_gop12.u64")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "RCCL all-gather of 128 chain hashes over 1 device(s): ok; parity: every stream equals the reference decoder" in p.stdout
    assert '"metric": "MPEG-1 352x192 frames/s"' in p.stdout
