#!/usr/bin/env python3
"""Per-picture frame hashes of the bench / BASELINE workloads, produced by RUNNING THE UNMODIFIED
REFERENCE (oracle/_ref/efx_ref_decode, built by `make ref` from /root/reference) on every stream.
Run in the build container only:

    make ref gen && python tests/golden/make_bench_golden.py

Writes (little-endian uint64, FNV-1a-64 of each pushed frame incl. the final flush_picture(1)):
  bench_gop12.u64    [8192][12]  SURVEY.md 8d config 3 / 5: stream ids 0..8191, GOP(12) = I + 11 P, flags 0
  bench_ionly.u64    [256][8]    SURVEY.md 8d config 2: stream ids 0..255, 8 I pictures (FLAG_I_ONLY)
  bench_wide1500k.u64 [1024][12] the service's stream shape: 5 slices per picture, ~6.25 kB per picture
                                 (FLAG_WIDE_SLICES | FLAG_RATE_1500K), ids 0..1023
  bench_wide1500k_p36.u64 [1024][36], bench_wide1500k_p72.u64 [1024][72]  the same shape as streams of 36 / 72 pictures = three /
                                 six GOP(12) (round 6: bench.py's pictures-per-call curve; the generator's rate control looks at
                                 the whole stream, so a shorter stream is not a prefix of a longer one)
bench.py compares every stream of its shard against these before the timed region; the -m gpu tests
compare every stream of BASELINE configs[1] / configs[2]; a CPU test pins the C restatement to a sample.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SETS = [("bench_gop12.u64", 8192, 12, 12, 0), ("bench_ionly.u64", 256, 8, 12, 1), ("bench_wide1500k.u64", 1024, 12, 12, 4 | 32),
        ("bench_wide1500k_p36.u64", 1024, 36, 12, 4 | 32), ("bench_wide1500k_p72.u64", 1024, 72, 12, 4 | 32)]
CHUNK = 64


def work(job):
    import oracle
    from espflix_amd import gen
    first, n, pictures, gop, flags = job
    b = gen.Batch(first, n, pictures, gop, flags, 1)
    out = np.zeros((n, pictures), dtype=np.uint64)
    for i in range(n):
        h, pts, _ = oracle.ref_decode(b.ts(i), flush_last=True)
        assert len(h) == pictures, (first + i, len(h))
        assert list(pts) == [129003 + 3003 * p for p in range(pictures)]
        out[i] = h
    return first, out


def main():
    import oracle
    assert oracle.have_ref(), "build oracle/_ref first (make ref)"
    only = sys.argv[1:]
    with mp.Pool(os.cpu_count() or 1) as pool:
        for name, n, pictures, gop, flags in SETS:
            if only and name not in only:
                continue
            jobs = [(f, min(CHUNK, n - f), pictures, gop, flags) for f in range(0, n, CHUNK)]
            table = np.zeros((n, pictures), dtype="<u8")
            for first, part in pool.imap_unordered(work, jobs):
                table[first:first + part.shape[0]] = part
            table.tofile(os.path.join(HERE, name))
            print(name, table.shape, f"xor {int(np.bitwise_xor.reduce(table.reshape(-1))):016x}")


if __name__ == "__main__":
    main()
