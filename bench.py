#!/usr/bin/env python3
"""bench.py -- MPEG-1 352x192 decode throughput of the MI355X hot path (BASELINE.json metric).

One "step" = one efx_decode() pass over the whole resident batch: start-code index, VLC parse,
dequantisation, IDCT, half-pel motion compensation and strip-layout store of every picture of
every stream.  Workload at N = 1 (BASELINE.json configs[2], the per-GPU shard of configs[4]):
1024 synthetic 352x192 streams x one GOP(12) = I + 11 P pictures,
bitstreams already resident in HBM when the timed region starts.  At N > 1 every rank decodes its
own 1024 streams (ids rank*1024 ...): weak scaling, no collective on the data path; RCCL is only
used for the barrier, the max-over-ranks time and the checksum-of-checksums report.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel's algorithmic GB/s vs the 8 TB/s HBM peak (HIP-event stage times
                measured inside the timed region on the decode stream)
  cpu_baseline  the unmodified reference decoder (oracle/_ref, kind "reference") -- or the C
                restatement (kind "port") when the reference build is absent -- timed on this
                box's host cores over the same TS-wrapped streams (N = 1, rank 0 only)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_BYTES = 101376
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic(kernel, S, P):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, separate runs, gfx950 correction applied by tools/summarize_profiles.py).  PMC
    counters cannot be collected from inside this process, so the figure is the one measured
    on this exact workload (1024 streams x GOP 12) and None for any other."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_summary.json")
    if (S, P) != (1024, 12) or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["kernels"].get(kernel, {}).get("hbm_traffic_bytes")


def algorithmic_bytes(es_bytes: int, n_i: int, n_p: int) -> int:
    """I picture = B + 101376, P picture = B + 202752."""
    return es_bytes + n_i * FRAME_BYTES + n_p * 2 * FRAME_BYTES


def usable_cores() -> int:
    """Host cores this process may actually use: CPU count, affinity mask and cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(batch, n_streams: int, n_pictures: int, budget_streams: int):
    """Time the reference decoder on this host over the first `budget_streams` streams."""
    cores = usable_cores()
    n = min(n_streams, budget_streams)
    ref = os.path.join(ROOT, "oracle", "_ref", "efx_ref_decode")
    sample = (f"first {n} of the {n_streams} streams x {n_pictures} pictures, TS-wrapped (PID 0x100), {cores} worker "
              f"processes = usable host cores (os.cpu_count() {os.cpu_count()}, cgroup/affinity limit {cores})")
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as td:
            lst = os.path.join(td, "list.txt")
            with open(lst, "w") as f:
                for i in range(n):
                    path = os.path.join(td, f"{i}.ts")
                    batch.ts(i).tofile(path)
                    f.write(path + "\n")
            # aim at ~5 s of wall time on all cores (the reference does ~2-4 k frames/s/core)
            repeat = max(1, int(5.0 * cores * 3000 / (n * n_pictures)))
            try:
                p = subprocess.run([ref, "bench", str(cores), lst, str(repeat)], stderr=subprocess.PIPE,
                                   stdout=subprocess.DEVNULL, text=True, timeout=150)
                err = p.stderr
            except subprocess.TimeoutExpired:
                err = ""
        line = [l for l in err.splitlines() if l.startswith("BENCH")]
        if line:
            kv = dict(x.split("=") for x in line[0].split()[1:])
            return {"value": int(kv["pictures"]) / float(kv["seconds"]), "unit": "frames/s", "cores": int(kv["workers"]),
                    "kind": "reference", "sample": sample + f", each worker replays its share {kv['repeat']}x "
                    f"({kv['pictures']} pictures in {float(kv['seconds']):.2f} s, {kv['failed']} plays failed)"}
    # fall back to the C restatement, one process per core
    import multiprocessing as mp
    blobs = [batch.ts(i) for i in range(n)]
    t0 = time.perf_counter()
    with mp.Pool(cores) as pool:
        counts = pool.map(_port_decode, blobs, chunksize=max(1, n // (4 * cores)))
    dt = time.perf_counter() - t0
    return {"value": sum(counts) / dt, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample}


def _port_decode(ts):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    n, _, _, _ = oracle.decode(ts, 1, flush_last=True, max_frames=64)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU")
    ap.add_argument("--pictures", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="synchronise after every step (no cross-step pipelining)")
    args = ap.parse_args()

    import torch
    import espflix_amd as efx
    from espflix_amd import gen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: espflix_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from espflix_amd import dist as edist
    first, S = edist.shard(rank, world, args.streams)
    P = args.pictures
    threads = max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    t_gen = time.perf_counter()
    batch = gen.Batch(first, S, P, 12, 0, threads)
    streams = batch.all_es()
    t_gen = time.perf_counter() - t_gen
    es_bytes = int(sum(s.size for s in streams))
    n_i = S * ((P + 11) // 12)
    n_p = S * P - n_i

    dec = efx.Decoder(max_streams=S, max_pictures=P, ring_depth=2, device=local_rank, max_stream_bytes=es_bytes + 64 * S)
    dec.upload(streams, efx.FORMAT_ES)  # bitstreams resident in HBM from here on
    dec.set_timing(True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):  # same enqueue pattern as the timed steps
        dec.decode(sync=args.no_overlap)
    dec.sync()
    barrier()
    dec.set_timing(True)  # new averaging window: the stage times below cover exactly the timed steps
    t0 = time.perf_counter()
    # Steps are enqueued back to back: libefx runs the parse half of step k+1 (parse stream)
    # while the reconstruction half of step k is still on the GPU (double-buffered hand-over),
    # exactly what a service decoding batch after batch does.  Every step still performs the
    # complete decode of the whole batch; all K steps finish inside the timed region.
    for _ in range(args.steps):
        dec.decode(sync=args.no_overlap)
    dec.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    # stage times: HIP events on the kernels' own streams, mean over the timed steps (the last 64)
    t = dec.timing()
    stage = np.array([t.index_ms, t.parse_ms, t.recon_ms]) * args.steps
    elapsed = edist.max_over_ranks(elapsed, dist, "cuda")

    # verification riding along: every stream decoded all its pictures with a clean status, and
    # a checksum of the per-stream frame checksums (deterministic for a given shard)
    t = dec.timing()
    assert t.pictures == S * P, (t.pictures, S * P)
    bad = [i for i in range(S) if dec.stream_status(i) != 0]
    assert not bad, f"streams with non-zero status: {bad[:8]}"
    hashes = dec.frame_hashes()
    csum = edist.xor_over_ranks(edist.frame_checksum(hashes), dist, "cuda", world)
    n_coefs = t.coefficients

    # outside the timed region: the same stages one call at a time (no overlap between calls), for
    # the uncontended per-kernel figures quoted next to the timed-region ones
    dec.set_timing(True)
    for _ in range(3):
        dec.decode(sync=True)
    ts = dec.timing()
    serial_ms = [ts.index_ms, ts.parse_ms, ts.recon_ms]

    if rank == 0:
        frames = world * S * P * args.steps
        value = frames / elapsed
        stage_ms = stage / args.steps
        names = ["k_index(+scan,emit)", "k_parse", "k_recon x%d" % P]
        alg = algorithmic_bytes(es_bytes, n_i, n_p)
        # The two halves take about the same GPU time (k_parse 1 launch, k_recon P launches).  The
        # roofline object is k_recon's: it is the kernel that moves SURVEY 8d's algorithmic bytes (the
        # frames); k_parse, which only reads the bitstream and writes 4 B per coefficient + 16 B per
        # macroblock, is reported next to it.
        k = 2
        launches = [1, 1, P]
        alg_launch = alg / P
        dur_s = stage_ms[k] / 1e3 / launches[k]
        achieved = alg_launch / dur_s / 1e9
        parse_bytes = es_bytes + 4 * n_coefs + 16 * S * P * 264
        traffic = pmc_traffic("efx::k_recon", S, P)
        out = {
            "metric": "MPEG-1 352x192 frames/s", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": f"{S} streams/GPU x GOP(12) I+11P, 352x192 MPEG-1 ES resident in HBM, "
                                   f"ids {first}..{first + S - 1} per rank (BASELINE configs[2]; shard of configs[4])",
                       "streams_per_gpu": S, "pictures_per_stream": P, "es_bytes_per_gpu": es_bytes,
                       "mean_bytes_per_picture": es_bytes / (S * P), "parallelism": f"stream-partition x{world}",
                       "coefficients_per_gpu": int(n_coefs),
                       "ring_depth": 2},
            "roofline": {"bound": "hbm", "kernel": names[k], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_launch, "avg_launch_ms": dur_s * 1e3,
                         "whole_step_achieved_GBs": alg / (elapsed / args.steps) / 1e9,
                         "stage_ms": dict(zip(names, [float(x) for x in stage_ms])),
                         "timed_calls_averaged": t.timed_calls,
                         "note": "stage_ms / avg_launch_ms are means over the timed steps, where the parse half of "
                                 "later steps shares the GPU with k_recon; serial_* = the same stages one call at a "
                                 "time, measured after the timed region",
                         "serial_stage_ms": dict(zip(names, [float(x) for x in serial_ms])),
                         "serial_frac": alg / P / (serial_ms[2] / P / 1e3) / 1e9 / HBM_PEAK_GBS,
                         "k_parse": {"algorithmic_bytes_per_launch": parse_bytes, "avg_launch_ms": float(stage_ms[1]),
                                     "achieved": parse_bytes / (stage_ms[1] / 1e3) / 1e9,
                                     "serial_launch_ms": float(serial_ms[1]),
                                     "traffic": pmc_traffic("efx::k_parse", S, P),
                                     "bound": "serial symbol chains (VALU issue), not bandwidth"}},
            "checksum_of_checksums": f"{csum:016x}",
            "gen_seconds": t_gen,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, S, P, 1024)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    dec.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
