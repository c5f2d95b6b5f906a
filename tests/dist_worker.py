"""Worker for tests/test_dist_gloo.py: one rank of a world_size-2 gloo job running bench.py's OWN
control flow (bench.run: partition, parity gate against the golden table, timed region with
barriers, max-over-ranks time, all-gather of per-stream chain hashes, counter sums, fixed-batch
partition, report) with the CPU oracle standing in for the GPU decoder object."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleDecoder:
    """espflix_amd.Decoder's interface as bench.py uses it, computed by the test oracle."""

    def __init__(self, max_streams, max_pictures, ring_depth, max_stream_bytes):
        self.P, self.D = max_pictures, max(2, ring_depth)
        self.calls = 0

    def upload(self, streams, fmt):
        import oracle
        self.tab = []
        for s in streams:
            n, h, _, _ = oracle.decode(s, fmt, flush_last=True, max_frames=self.P)
            self.tab.append(h[:n])
        self.S = len(streams)

    def prepare_upload(self, streams):
        return streams

    def upload_prepared(self, prepared, fmt=0):
        if not hasattr(self, "tab"):
            self.upload(prepared, fmt)

    def decode(self, sync=True, first_picture=0):
        self.calls += 1

    def sync(self):
        pass

    def picture_slot(self, p, stream=0):
        return (p + 1) % self.D

    def picture_count(self, i):
        return len(self.tab[i])

    def stream_status(self, i):
        return 0

    def frame_hashes(self):
        out = np.zeros((self.S, self.D), dtype=np.uint64)
        for i, h in enumerate(self.tab):
            for p in range(len(h)):      # later pictures overwrite earlier ones, as in the ring
                out[i, (p + 1) % self.D] = h[p]
        return out

    def set_timing(self, on):
        self.calls = 0

    def timing(self):
        return types.SimpleNamespace(index_ms=0.1, parse_ms=1.0, recon_ms=1.0, total_ms=2.1, pictures=sum(len(h) for h in self.tab),
                                     coefficients=1000 * self.S, timed_calls=self.calls, slices=0, es_bytes=0, groups=2)

    def close(self):
        pass


def main():
    import torch.distributed as dist
    import bench

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    args = bench.parse_args(sys.argv[1:])
    job = bench.Job(rank, world, dist if world > 1 else None, "cpu", OracleDecoder)
    out = bench.run(job, args)
    if rank == 0:
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
