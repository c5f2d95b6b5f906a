"""Worker for tests/test_dist_gloo.py: one rank of a world_size-2 gloo job running the bench's
partition + reduction logic, with the CPU oracle standing in for the GPU decode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from espflix_amd import dist as edist
    from espflix_amd import gen
    import oracle

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    per = int(sys.argv[1])
    first, n = edist.shard(rank, world, per)
    b = gen.Batch(first, n, 4, 12, 0, 1)
    hashes = np.concatenate([oracle.decode(b.es(i), 0)[1] for i in range(n)])
    dist.barrier()
    elapsed = edist.max_over_ranks(1.0 + rank, dist, "cpu")          # slowest rank defines the time
    csum = edist.xor_over_ranks(edist.frame_checksum(hashes), dist, "cpu", world)
    if rank == 0:
        print(f"RESULT {elapsed} {csum:016x} {world * n * 4}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
