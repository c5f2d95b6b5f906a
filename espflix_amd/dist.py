"""Multi-GPU plumbing for the stream-partitioned decode.

Streams are fully independent, so scaling is a pure partition: rank r owns stream ids
[r * per_gpu, (r + 1) * per_gpu) and no collective touches the data path.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the
bracketing barrier, the max-over-ranks wall time, the all-gather of per-stream frame-chain hashes
(8 bytes per stream) and the counter sums.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def shard(rank: int, world: int, per_gpu: int):
    """(first stream id, stream count) owned by `rank`; contiguous blocks, weak scaling."""
    if not (0 <= rank < world) or per_gpu <= 0:
        raise ValueError("bad shard arguments")
    return rank * per_gpu, per_gpu


def shard_fixed(rank: int, world: int, total: int):
    """[lo, hi) of a FIXED batch of `total` streams owned by `rank`: stream k lives on rank
    floor(k * world / total), i.e. lo = ceil(rank * total / world)."""
    if not (0 <= rank < world) or total <= 0:
        raise ValueError("bad shard arguments")
    lo = -(-rank * total // world)
    hi = -(-(rank + 1) * total // world)
    return lo, hi


def frame_checksum(hashes: np.ndarray) -> int:
    """Order-independent 64-bit digest of per-frame hashes (xor of multiplicatively mixed values)."""
    h = np.ascontiguousarray(hashes, dtype=np.uint64).reshape(-1)
    with np.errstate(over="ignore"):
        return int(np.bitwise_xor.reduce(h * GOLDEN)) if h.size else 0


def max_over_ranks(value: float, dist, device) -> float:
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def xor_over_ranks(csum: int, dist, device, world: int) -> int:
    """all_gather of the per-rank digests (8 bytes per rank), xor-combined."""
    import torch
    if dist is None:
        return csum
    lo = torch.tensor([csum & 0x7FFFFFFF, (csum >> 31) & 0x7FFFFFFF, csum >> 62], dtype=torch.int64, device=device)
    out = [torch.zeros_like(lo) for _ in range(world)]
    dist.all_gather(out, lo)
    total = 0
    for o in out:
        a, b, c = (int(x) for x in o.tolist())
        total ^= a | (b << 31) | (c << 62)
    return total


def gather_u64(values: np.ndarray, dist, device, world: int) -> np.ndarray:
    """all_gather of one uint64 array per rank (per-stream chain hashes: 8 bytes per stream, 64 KiB for
    the 8192-stream job), concatenated in rank order.  Ranks may hold different counts."""
    import torch
    v = np.ascontiguousarray(values, dtype=np.uint64)
    if dist is None:
        return v.copy()
    n = torch.tensor([v.size], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    buf = np.zeros(cap, dtype=np.int64)
    buf[: v.size] = v.view(np.int64)
    mine = torch.from_numpy(buf).to(device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)]).view(np.uint64)


def sum_over_ranks(values, dist, device):
    """all_reduce(sum) of a few integer counters (pictures, bytes, coefficients)."""
    import torch
    if dist is None:
        return [int(x) for x in values]
    t = torch.tensor([int(x) for x in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def bind_rank_to_device_node(pci_bus_id: str) -> dict:
    """One process per GPU: bind this rank (its main thread -- threads it starts later inherit the mask, libefx's staging
    copy threads included) to the CPUs of the NUMA node its device hangs off, BEFORE the decoder context allocates its pinned
    staging memory (first touch puts it on that node).  At 8 ranks per node the PCIe-inclusive path moves ~35 GB/s per
    rank through host memory; a rank on the wrong socket pays the inter-socket link for every byte.  Returns what was done;
    a no-op where sysfs knows no node for the device or none of the node's CPUs belong to this process."""
    import espflix_amd as efx
    node = efx.numa_node_of_pci(pci_bus_id)
    return {"node": node, "cpus_bound": efx.numa_bind_thread(node) if node >= 0 else 0}
