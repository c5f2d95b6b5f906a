#!/usr/bin/env python3
"""bench.py -- MPEG-1 352x192 decode throughput of the MI355X hot path (BASELINE.json metric).

One "step" = one efx_decode() pass over the whole resident batch: start-code index, VLC parse,
dequantisation, IDCT, half-pel motion compensation and strip-layout store of every picture of
every stream.  Workload (BASELINE.json configs[2], the per-GPU shard of configs[4]): 1024 synthetic
352x192 streams per GPU x one GOP(12) = I + 11 P pictures, bitstreams
already resident in HBM when the timed region starts.  Rank r of N decodes stream ids
[r*1024, (r+1)*1024): weak scaling, no collective on the data path -- at N = 8 this IS configs[4]
(8192 streams, stream k on rank floor(k*8/8192)).  RCCL is used for the barriers, the
max-over-ranks time, the all-gather of per-stream frame-chain hashes and the counter sums.

    python bench.py --gpus N --steps K --warmup W       N > 1 without a torchrun environment re-launches
                                                         itself under torch.distributed.run (one rank per GPU)

Parity gate: BEFORE the timed region every stream of the shard is decoded
with all pictures kept and every picture's frame hash is compared with tests/golden/bench_*.u64 --
the output of the unmodified reference decoder on the same streams (tests/golden/make_bench_golden.py).
After the timed region the two frames left in the double buffer are compared again, the per-stream
chain hashes are all-gathered and rank 0 checks the whole job against the golden table.  A mismatch
aborts the run: no number is printed for a decoder that differs from the reference.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline          dominant kernel's algorithmic GB/s vs the 8 TB/s HBM peak (HIP-event stage times
                    measured inside the timed region on the kernels' own streams)
  cpu_baseline      the unmodified reference decoder (oracle/_ref, kind "reference") -- or the C
                    restatement (kind "port") when the reference build is absent -- timed on this
                    box's host cores over the same TS-wrapped streams (rank 0, every N)
  fixed_batch_8192  SURVEY.md 8d config 5 as written: the fixed 8192-stream batch, stream k on rank
                    floor(k*N/8192) (strong scaling), timed after the primary region
  other_workloads   the service's real stream shape (5 slices per picture, ~6.25 kB per picture) and the
                    reference's embedded clip replicated x1024, with stage times (N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_BYTES = 101376
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FIXED_BATCH = 8192
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
FNV_BASIS, FNV_PRIME, M64 = 0xCBF29CE484222325, 0x100000001B3, (1 << 64) - 1

# name -> (generator flags, golden table, rows in the table, pictures per row)
WORKLOADS = {
    "gop12": (0, "bench_gop12.u64", 8192, 12),
    "wide1500k": (4 | 32, "bench_wide1500k.u64", 1024, 12),
    # the same shape as streams of 36 and 72 pictures (three / six GOPs), decoded in one call: the reference's hashes of those streams
    "wide1500k_p36": (4 | 32, "bench_wide1500k_p36.u64", 1024, 36),
    "wide1500k_p72": (4 | 32, "bench_wide1500k_p72.u64", 1024, 72),
}


# ONE convention for vector-instruction issue (round 6; profiles/r6_valu_rates.md, measured by tools/exp/valu_rates.hip):
# a SIMD issues a wave64 instruction of the FULL-rate class (v_add / v_sub / v_and / v_or / v_xor / v_mov / v_lshrrev / v_ashrrev /
# v_bitop3 / v_fma_f32) every 2.2 cycles and of the HALF-rate class (everything else this path uses: 24-bit multiplies, byte
# permutes, v_lerp_u8, packed 16-bit, v_bfi, v_cndmask, v_cmp, v_lshlrev, v_min / v_max, DPP) every 4.2; a wave that is ALONE on
# its SIMD issues one instruction per 4.4 cycles whatever it is.  profiles/r6_issue_floor.json (tools/issue_floor.py) holds the
# mix-weighted cycles per instruction of every kernel and, with rocprofv3's SQ_INSTS_VALU, its issue floor per launch.
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
PDM_VALU_PER_STEP = 9.3           # VALU instructions per delta-sigma step and lane in k_pdm's main loop (hipcc -S: 2382 per 8 samples = 256
                                  # steps; 8 in the step itself, the rest is the sample's low-pass and the word assembly); 10.2 in round 5


def issue_model():
    """profiles/r6_issue_floor.json (kernel -> cycles per VALU instruction, dynamic VALU per launch), or {}."""
    path = os.path.join(ROOT, "profiles", "r6_issue_floor.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get("kernels", {})


def log(msg):
    sys.stderr.write(f"[bench rank {os.environ.get('RANK', '0')}] {msg}\n")
    sys.stderr.flush()


def _kernel_source_files():
    """The kernel sources proper (espflix_amd/csrc: the k_*.hip files, the headers and the table builder -- not the host side,
    efx_api.hip / efx_multi.cpp)."""
    d = os.path.join(ROOT, "espflix_amd", "csrc")
    return [(n, os.path.join(d, n)) for n in sorted(os.listdir(d))
            if (n.startswith("k_") and n.endswith(".hip")) or n.endswith(".h") or n == "efx_tables.cpp"]


def kernel_source_digests() -> dict:
    """SHA-1 per kernel source file: what a PMC summary was measured on (tools/summarize_profiles.py), and what this run
    executes."""
    import hashlib
    out = {}
    for name, path in _kernel_source_files():
        with open(path, "rb") as f:
            out[name] = hashlib.sha1(f.read()).hexdigest()[:12]
    return out


def kernel_sources_digest() -> str:
    """One SHA-1 over all of them."""
    import hashlib
    h = hashlib.sha1()
    for name, path in _kernel_source_files():
        with open(path, "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


# the file a kernel of the summaries lives in (beside the headers and the tables, which every kernel includes)
_KERNEL_FILE = {"k_recon": "k_recon.hip", "k_recon_all": "k_recon.hip", "k_parse": "k_parse.hip", "k_composite": "k_video.hip",
                "k_pdm": "k_video.hip", "k_demux": "k_demux.hip", "k_sbc": "k_sbc.hip", "k_index": "k_index.hip",
                "k_slice_scan": "k_index.hip", "k_slice_emit": "k_index.hip", "k_advance": "k_index.hip"}


def pmc_traffic(kernel, S, P, key="kernels"):
    """(HBM bytes per launch of `kernel`, provenance) from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE, separate runs, gfx950 correction applied by
    tools/summarize_profiles.py).  PMC counters cannot be collected from inside this process, so
    the figure is the one measured on this exact workload (1024 streams x GOP 12), None otherwise.
    The summary records a digest per kernel source file it was measured on; when the kernel's own file, a header or the
    table builder has changed since, the provenance says so and a warning goes to stderr."""
    for name in ("r6_pmc_summary.json", "r5_pmc_summary.json", "r4_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json", "r1_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        if (S, P) != (1024, 12) or not os.path.exists(path):
            continue
        with open(path) as f:
            doc = json.load(f)
        v = doc.get(key, {}).get(kernel, {}).get("hbm_traffic_bytes")
        if v is None:
            continue
        base = kernel.split("::")[-1].split(":")[0]
        own = next((f for k, f in sorted(_KERNEL_FILE.items(), key=lambda kv: -len(kv[0])) if base.startswith(k)), None)
        then, now = doc.get("kernel_source_files"), kernel_source_digests()
        if then is not None:
            moved = [f for f in now if (f == own or not f.endswith(".hip")) and then.get(f) != now[f]]
            measured_on, this_run = {f: then.get(f) for f in moved}, {f: now[f] for f in moved}
        else:  # (summaries of earlier rounds: one digest over everything)
            measured_on, this_run = doc.get("kernel_sources_digest"), kernel_sources_digest()
            moved = [] if measured_on == this_run else ["(all kernel sources)"]
        if moved:
            log(f"warning: {name} was measured before {', '.join(moved)} changed: roofline.traffic of {kernel} may be out of date "
                f"(re-run tools/collect_profiles.sh)")
        # (round-5 verdict: the two fields were empty whenever nothing had changed) what the passes were measured ON and what
        # this run executes, always: the digest over all kernel sources, the collection's own note, and the files that differ
        return v, {"file": "profiles/" + name, "stale": bool(moved), "changed_since": moved,
                   "measured_on": {"kernel_sources_digest": doc.get("kernel_sources_digest"), "collected": doc.get("note", ""), "files": measured_on},
                   "this_run": {"kernel_sources_digest": kernel_sources_digest(), "files": this_run},
                   "note": "separate rocprofv3 --pmc passes of this command (tools/collect_profiles.sh), not this run"}
    return None, None


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


def chain_hashes(table: np.ndarray) -> np.ndarray:
    """Per-stream frame-chain hash: FNV-1a-64 over the 8 little-endian bytes of each
    successive frame hash.  table: [streams][pictures] uint64."""
    out = np.empty(table.shape[0], dtype=np.uint64)
    for i, row in enumerate(np.ascontiguousarray(table, dtype="<u8")):
        h = FNV_BASIS
        for b in row.tobytes():
            h = ((h ^ b) * FNV_PRIME) & M64
        out[i] = h
    return out


def load_golden(workload: str):
    flags, fname, rows, pics = WORKLOADS[workload][:4]
    cols = WORKLOADS[workload][4] if len(WORKLOADS[workload]) > 4 else pics  # (columns of the table: pictures of its streams)
    path = os.path.join(GOLDEN_DIR, fname)
    if not os.path.exists(path):
        raise SystemExit(f"{path} is missing: the bench refuses to time a decoder it cannot check against the reference")
    return np.ascontiguousarray(np.fromfile(path, dtype="<u8").reshape(rows, cols)[:, :pics])


def cpu_baseline(batch, n_streams: int, n_pictures: int, budget_streams: int, seconds: float = 5.0):
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
            # aim at ~`seconds` of wall time on all cores (the reference does ~2-4 k frames/s/core)
            repeat = max(1, int(seconds * cores * 3000 / (n * n_pictures)))
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


def _bench_line(err: str) -> dict:
    line = [l for l in err.splitlines() if l.startswith("BENCH")]
    return dict(x.split("=") for x in line[0].split()[1:]) if line else {}


def cpu_baseline_o3(batch, n_streams: int, n_pictures: int, budget_streams: int, seconds: float):
    """SURVEY 8d's second CPU line: the same unmodified reference sources at -O3 with AVX2 (oracle/Makefile: the binary
    is built away from this host, so x86-64-v3 instead of -march=native; skipped when the host lacks AVX2)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "efx_ref_decode_o3")
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if not os.path.exists(ref) or not all(f" {x}" in flags for x in ("avx2", "bmi2", "fma")):
        return None
    cores, n = usable_cores(), min(n_streams, budget_streams)
    with tempfile.TemporaryDirectory() as td:
        lst = os.path.join(td, "list.txt")
        with open(lst, "w") as f:
            for i in range(n):
                path = os.path.join(td, f"{i}.ts")
                batch.ts(i).tofile(path)
                f.write(path + "\n")
        repeat = max(1, int(seconds * cores * 3000 / (n * n_pictures)))
        try:
            p = subprocess.run([ref, "bench", str(cores), lst, str(repeat)], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL,
                               text=True, timeout=150)
        except subprocess.TimeoutExpired:
            return None
    kv = _bench_line(p.stderr)
    if not kv:
        return None
    return {"value": int(kv["pictures"]) / float(kv["seconds"]), "unit": "frames/s", "cores": int(kv["workers"]), "kind": "reference",
            "build": "-O3 -march=x86-64-v3 (AVX2): the reference sources unmodified, oracle/Makefile",
            "sample": f"first {n} of the {n_streams} streams x {n_pictures} pictures, {kv['workers']} worker processes, each replays its "
                      f"share {kv['repeat']}x ({kv['pictures']} pictures in {float(kv['seconds']):.2f} s, {kv['failed']} plays failed)"}


def _event_ms(torch, stream, fn, reps):
    """Mean duration of fn() in ms over `reps` back-to-back calls, measured with HIP events ON the stream the kernels
    are launched on (the decoder context was created on this torch stream)."""
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            fn(i)
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def run_video_out(job, args, S):
    """BASELINE configs[3] and the SURVEY 8f rows in the driver-timed line: composite fields (NTSC, PAL), the PDM
    modulator, TS demux and SBC decode over a batch of S streams -- each leg first checked against the goldens the
    unmodified reference produced (tests/golden/golden.json), then timed with HIP events on the kernels' own stream;
    next to them the reference's own video_isr() / write_pcm_16() timed on this box's host cores."""
    import torch
    import espflix_amd as efx
    from espflix_amd import gen
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common  # shared deterministic inputs of the golden vectors (no oracle involved)
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        golden = json.load(f)
    fnv = lambda a: f"{gen.fnv1a64(a):016x}"
    stream = torch.cuda.Stream()
    mk = lambda n, p, d, **kw: efx.Decoder(max_streams=n, max_pictures=p, ring_depth=d, device=torch.cuda.current_device(),
                                           hip_stream=stream.cuda_stream, **kw)
    out = {"streams": S, "timing": "HIP events on the stream the kernels run on, mean over back-to-back launches"}

    # ---- composite fields -----------------------------------------------------------------------------------------
    dec = mk(S, 2, 2)
    lcg = common.lcg_frames()
    dec.upload_frame(0, 0, lcg[:FRAME_BYTES])
    for ntsc in (True, False):
        vp = efx.video_params(ntsc)
        n = vp["line_width"] * vp["line_count"]
        one = dec.alloc(n * 2)
        got = []
        for fc in range(3):
            dec.composite_fields(one, 0, 1, 0, ntsc, fc)
            dec.sync()
            got.append(fnv(one.download(np.uint16, n)))
        one.free()
        if got != golden["composite"]["lcg:" + ("ntsc" if ntsc else "pal")]:
            raise SystemExit("parity gate (video_out): composite fields differ from the reference's video_isr()")
    b = gen.Batch(0, S, 2, 12, 0, max(1, usable_cores()))
    dec.upload(b.all_es(), 0)
    dec.decode()
    slot = dec.picture_slot(1)
    out["composite"] = {}
    for ntsc in (True, False):
        std = "ntsc" if ntsc else "pal"
        vp = efx.video_params(ntsc)
        n = vp["line_width"] * vp["line_count"]
        dst = dec.alloc(S * n * 2)
        for i in range(3):
            dec.composite_fields(dst, 0, S, slot, ntsc, i)
        dec.sync()
        ms = _event_ms(torch, stream, lambda i: dec.composite_fields(dst, 0, S, slot, ntsc, i), 50)
        alg = S * (FRAME_BYTES + n * 2)  # the frame read once + every sample of the field written
        traffic, src = pmc_traffic("efx::k_composite:" + std, 1024, 12, key="video_kernels") if S == 1024 else (None, None)
        out["composite"][std] = {"fields_per_s": S / ms * 1e3, "ms_per_launch": ms,
                                 "roofline": {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": alg / ms / 1e6 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                                              "algorithmic_bytes_per_launch": alg}}
        dst.free()
    dec.close()

    # ---- PDM: golden check, then stream-seconds per second at S and at stream counts that fill the chip ----------------
    dec = mk(1, 1, 2)
    pcm = common.pdm_pcm(0, 40)
    d_pcm, d_state, d_out = dec.alloc(pcm.nbytes), dec.alloc(12), dec.alloc(pcm.size * 4)
    d_pcm.upload(pcm)
    d_state.upload(np.zeros(3, dtype=np.int32))
    dec.pdm(1, d_pcm, pcm.size, d_state, d_out)
    dec.sync()
    if fnv(d_out.download(np.uint16, 2 * pcm.size)) != golden["pdm"]["sine220"]:
        raise SystemExit("parity gate (video_out): PDM words differ from the reference's pdm_second_order()")
    for bfr in (d_pcm, d_state, d_out):
        bfr.free()
    pdm_cpi = issue_model().get("k_pdm", {}).get("cycles_per_valu", 3.0)
    out["pdm"] = {"what": "stream-seconds of 48 kHz audio modulated per second (32 delta-sigma steps per sample, one lane per stream: "
                          "the recurrence is serial, so the rate grows with the stream count until every SIMD holds waves -- 1024 "
                          "streams are 16 waves on a chip of 1024 SIMDs: an occupancy artefact, not a property of the kernel)",
                  "bound": "valu_issue",
                  "issue_note": "issue_frac = streams / 64 waves x samples x 32 steps x %.1f vector instructions per step x %.2f cycles per "
                                "instruction (the kernel's mix at the measured rates, profiles/r6_issue_floor.json) / (1024 SIMDs x 2.4 GHz x "
                                "time); at 1024 streams the 16 waves sit alone on 16 of the 1024 SIMDs and issue one instruction per ~4 cycles "
                                "whatever its class: the time is samples x 32 x instructions per step x that interval" % (PDM_VALU_PER_STEP, pdm_cpi),
                  "by_streams": {}}
    for n_streams, samples in ((S, 48000), (16 * S, 9600), (64 * S, 4800), (256 * S, 2400)):
        d_pcm, d_state, d_out = dec.alloc(n_streams * samples * 2), dec.alloc(n_streams * 12), dec.alloc(n_streams * samples * 4)
        one = np.round(8000 * np.sin(2 * np.pi * 220 * np.arange(samples) / 48000)).astype(np.int16)
        d_pcm.upload(np.tile(one, n_streams))
        d_state.upload(np.zeros(n_streams * 3, dtype=np.int32))
        dec.pdm(n_streams, d_pcm, samples, d_state, d_out)
        dec.sync()
        ms = _event_ms(torch, stream, lambda i: dec.pdm(n_streams, d_pcm, samples, d_state, d_out), 3)
        alg = n_streams * samples * 6
        # the recurrence's roof is the integer VALU, not HBM: 32 delta-sigma steps per sample, PDM_VALU_PER_STEP VALU
        # instructions per step and lane in the kernel's 8-sample loop body (hipcc -S: 2.6 k v_* for 256 steps)
        wave_instr = (n_streams + 63) // 64 * samples * 32 * PDM_VALU_PER_STEP
        out["pdm"]["by_streams"][str(n_streams)] = {"samples_per_stream": samples, "ms_per_launch": ms,
                                                    "stream_seconds_per_s": n_streams * samples / 48000 / ms * 1e3,
                                                    "issue_frac": wave_instr * pdm_cpi / (SIMDS * CLOCK_HZ * ms * 1e-3),
                                                    "achieved_GBs": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS}
        for bfr in (d_pcm, d_state, d_out):
            bfr.free()
    dec.close()

    # ---- TS demux on the device ---------------------------------------------------------------------------
    b8 = gen.Batch(0, 8, 12, 12, 0)
    dec = mk(8, 12, 2)
    dec.upload([b8.ts(k) for k in range(8)], 1)
    for k in common.SYN_IDS:
        if fnv(np.frombuffer(dec.es(k), dtype=np.uint8)) != golden["synthetic"][f"0:{k}"]["es_fnv"]:
            raise SystemExit("parity gate (video_out): demultiplexed elementary stream differs")
    dec.close()
    bS = gen.Batch(0, S, 12, 12, 0, max(1, usable_cores()))
    ts = [bS.ts(k) for k in range(S)]
    es_bytes = sum(bS.es(k).size for k in range(S))
    dec = mk(S, 12, 2, max_stream_bytes=sum(x.size for x in ts) + 4096)
    dec.set_timing(True)
    best = None
    for _ in range(5):
        dec.upload(ts, 1)
        dec.decode()
        t = dec.timing()
        best = t.demux_ms if best is None else min(best, t.demux_ms)
    alg = t.ts_bytes + es_bytes
    out["demux"] = {"ms_per_launch": best, "ts_bytes": int(t.ts_bytes), "es_bytes": int(es_bytes), "ts_GB_per_s": t.ts_bytes / best / 1e6,
                    "roofline": {"bound": "hbm", "achieved": alg / best / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg / best / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": int(alg)}}
    dec.close()

    # ---- SBC audio decode ---------------------------------------------------------------------------------
    name, kw, nfr, _ = common.SBC_CASES[1]  # (a case without the probe call: the golden is the whole PCM)
    gfb = common.sbc_frame_bytes(kw["blocks"], 1, kw["bitpool"])
    fr = common.sbc_frames(common.seed_of(name), nfr, **kw)
    dec = mk(1, 1, 2)
    d_fr, d_st, d_pcm, d_cnt = dec.alloc(fr.size + 16), dec.alloc(efx.sbc_state_bytes()), dec.alloc(nfr * 256 * 2), dec.alloc(4)
    d_fr.upload(fr)
    d_st.upload(np.zeros(efx.sbc_state_bytes(), dtype=np.uint8))
    dec.sbc_decode(1, d_fr, (fr.size + 15) & ~15, gfb, nfr, d_st, d_pcm, nfr * 256, None, d_cnt)
    dec.sync()
    cnt = int(d_cnt.download(np.uint32, 1)[0])
    if fnv(d_pcm.download(np.int16, nfr * 256)[:cnt]) != golden["sbc"][name]:
        raise SystemExit("parity gate (video_out): SBC PCM differs from the reference's sbc_decoder()")
    for bfr in (d_fr, d_st, d_pcm, d_cnt):
        bfr.free()
    name, kw, _, _ = common.SBC_CASES[0]  # the service's own audio format: 48 kHz mono, 16 blocks, bitpool 28
    fb = common.sbc_frame_bytes(kw["blocks"], 1, kw["bitpool"])
    frames = 375  # one second of 48 kHz mono audio
    one = common.sbc_frames(1, frames, **kw)
    d_fr, d_st, d_pcm = dec.alloc(S * frames * fb + 1024), dec.alloc(S * efx.sbc_state_bytes()), dec.alloc(S * frames * 128 * 2)
    d_cnt = dec.alloc(S * 4)
    zeros = np.zeros(S * efx.sbc_state_bytes(), dtype=np.uint8)
    alg = S * frames * (fb + 256)
    out["sbc"] = {"frames_per_stream": frames,
                  "what": "stream-seconds of 48 kHz mono audio decoded per second; clean = every frame decodes (the regular frame-parallel kernel), "
                          "mixed = every fourth stream carries frames the reference rejects or that change the geometry (bad sync bytes, joint "
                          "stereo, 4-subband headers, other block counts / bitpools: common.sbc_mutate) -- resolved by k_sbc_plan, decoded by "
                          "k_sbc_gen and, chunk-wise where a run of frames is regular again, by the regular kernels"}
    rng = np.random.default_rng(5)
    dirty = [common.sbc_mutate(rng, one, fb, frames, hits=1 + i % 3) for i in range(16)]
    for mix in ("clean", "mixed"):
        batch = [dirty[(i // 4) % 16] if (mix == "mixed" and i % 4 == 0) else one for i in range(S)]
        d_fr.upload(np.concatenate(batch))
        pcm = {}
        for serial in (1, 0):  # the one-wave-per-stream kernel first: what the frame-parallel kernels must reproduce
            dec.set_option(efx.OPT_SBC_SERIAL, serial)
            d_st.upload(zeros)
            dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128, None, d_cnt)
            dec.sync()
            cnt = d_cnt.download(np.uint32, S)
            pcm[serial] = (cnt, d_pcm.download(np.int16, S * frames * 128).reshape(S, -1))
        same = np.array_equal(pcm[0][0], pcm[1][0]) and all(np.array_equal(pcm[0][1][i, :pcm[0][0][i]], pcm[1][1][i, :pcm[0][0][i]])
                                                             for i in range(0, S, 4))
        if not same:
            raise SystemExit("parity gate (video_out): the frame-parallel SBC kernels and the one-wave kernel differ (%s batch)" % mix)
        ms = _event_ms(torch, stream, lambda i: dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128), 5)
        out["sbc"][mix] = {"ms_per_call": ms, "stream_seconds_per_s": S / ms * 1e3,
                           "roofline": {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                                        "note": "VALU-bound (profiles/r5_sbc.md): 82 M vector instructions per call in k_sbc_par_mono"}}
    out["sbc"]["ms_per_launch"] = out["sbc"]["clean"]["ms_per_call"]
    out["sbc"]["stream_seconds_per_s"] = out["sbc"]["clean"]["stream_seconds_per_s"]
    out["sbc"]["roofline"] = out["sbc"]["clean"]["roofline"]
    dec.close()

    # ---- the reference's own video-out on this box's host cores --------------------------------------------------------------
    cores = usable_cores()
    cpu = {}
    rv, rp = os.path.join(ROOT, "oracle", "_ref", "efx_ref_video"), os.path.join(ROOT, "oracle", "_ref", "efx_ref_pdm")
    if os.path.exists(rv) and os.path.exists(rp) and not args.no_cpu_baseline:
        with tempfile.TemporaryDirectory() as td:
            fpath, ppath = os.path.join(td, "frames.bin"), os.path.join(td, "pcm.bin")
            lcg.tofile(fpath)
            common.pdm_pcm(0, 375).tofile(ppath)
            for ntsc in (1, 0):
                p = subprocess.run([rv, "bench", fpath, str(ntsc), "40000", str(cores)], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL,
                                   text=True, timeout=120)
                kv = _bench_line(p.stderr)
                if kv:
                    cpu["composite_" + ("ntsc" if ntsc else "pal")] = {
                        "value": int(kv["fields"]) / float(kv["seconds"]), "unit": "fields/s", "cores": int(kv["workers"]), "kind": "reference",
                        "sample": f"video_isr() (src/video.cpp, unmodified, -O2) over {kv['fields']} fields of the LCG frame, "
                                  f"{kv['workers']} worker processes, {float(kv['seconds']):.2f} s"}
            p = subprocess.run([rp, "bench", ppath, "800", str(cores)], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True,
                               timeout=120)
            kv = _bench_line(p.stderr)
            if kv:
                cpu["pdm"] = {"value": int(kv["samples"]) / 48000 / float(kv["seconds"]), "unit": "stream-seconds/s", "cores": int(kv["workers"]),
                              "kind": "reference",
                              "sample": f"write_pcm_16() / pdm_second_order() (espflix.ino:73-145, text lifted at build time, -O2) over "
                                        f"{kv['samples']} samples, {kv['workers']} worker processes, {float(kv['seconds']):.2f} s"}
    out["cpu_baseline_video"] = cpu or None
    return out


# ---------------------------------------------------------------------------------------------------
class Job:
    """Rank-local state of one bench run: torch.distributed handle + the decoder factory.
    `make_decoder(max_streams, max_pictures, ring_depth, max_stream_bytes)` returns an object with
    the espflix_amd.Decoder interface; the gloo tests substitute a CPU stand-in to run this exact
    control flow (partition, gate, gather, report) without a GPU."""

    def __init__(self, rank, world, dist, device, make_decoder, local_world=None):
        self.rank, self.world, self.dist, self.device, self.make_decoder = rank, world, dist, device, make_decoder
        self.local_world = local_world or world

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.device == "cuda":
            import torch
            torch.cuda.synchronize()


def gate_against_golden(dec_factory, streams, ids, golden, P, what):
    """Decode `streams` once with every picture kept and compare every frame hash with the golden
    table (rows = stream ids).  Returns the [n][P] table actually decoded."""
    S = len(streams)
    es_bytes = int(sum(s.size for s in streams))
    dec = dec_factory(S, P, P + 1, es_bytes + 64 * S)
    dec.upload(streams, 0)
    dec.decode()
    hashes = dec.frame_hashes()
    got = np.empty((S, P), dtype=np.uint64)
    for p in range(P):
        got[:, p] = hashes[:, dec.picture_slot(p)]
    counts = np.array([dec.picture_count(i) for i in range(S)])
    status = np.array([dec.stream_status(i) for i in range(S)])
    dec.close()
    if not (counts == P).all() or status.any():
        bad = np.nonzero((counts != P) | (status != 0))[0][:8]
        raise SystemExit(f"parity gate ({what}): streams {[int(ids[b]) for b in bad]} decoded {counts[bad]} pictures, status {status[bad]}")
    want = golden[np.asarray(ids) % golden.shape[0]]
    if not np.array_equal(got, want):
        bad = np.nonzero((got != want).any(axis=1))[0]
        raise SystemExit(f"parity gate ({what}): {bad.size} of {S} streams differ from the reference decoder, first ids "
                         f"{[int(ids[b]) for b in bad[:8]]}")
    return got


def timed_region(job, dec, steps, warmup, overlap=True):
    """W untimed + K timed efx_decode passes, barrier + synchronize on both sides, max over ranks."""
    from espflix_amd import dist as edist
    for _ in range(warmup):  # same enqueue pattern as the timed steps
        dec.decode(sync=not overlap)
    dec.sync()
    job.barrier()
    dec.set_timing(True)  # new averaging window: the stage times cover exactly the timed steps
    t0 = time.perf_counter()
    # Steps are enqueued back to back: libefx runs the parse half of step k+1 (parse stream)
    # while the reconstruction half of step k is still on the GPU (double-buffered hand-over),
    # exactly what a service decoding batch after batch does.  Every step still performs the
    # complete decode of the whole batch; all K steps finish inside the timed region.
    for _ in range(steps):
        dec.decode(sync=not overlap)
    dec.sync()
    job.barrier()
    elapsed = time.perf_counter() - t0
    return edist.max_over_ranks(elapsed, job.dist, job.device)


_GEN_CACHE = {}  # --soak: a leg is repeated with the same streams


def run_workload(job, args, workload, S, ids, gen_threads, steps, warmup, want_serial=True, sustained_steps=0):
    """Generate, gate, time one workload on this rank.  Returns a dict of rank-local + job results."""
    from espflix_amd import dist as edist
    from espflix_amd import gen
    flags, _, _, P = WORKLOADS[workload][:4]
    golden = load_golden(workload)
    t_gen = time.perf_counter()
    # ids are contiguous per rank (or wrapped for the small tables): generate in runs of consecutive ids
    key = (workload, S, int(ids[0]), int(ids[-1]))
    if key in _GEN_CACHE:
        batches, streams = _GEN_CACHE[key]
    else:
        batches, streams = [], []
        start = 0
        while start < S:
            end = start + 1
            while end < S and ids[end] == ids[end - 1] + 1:
                end += 1
            b = gen.Batch(int(ids[start]), end - start, P, 12, flags, gen_threads)
            batches.append((start, b))
            streams.extend(b.all_es())
            start = end
        if getattr(args, "soak", None):
            _GEN_CACHE[key] = (batches, streams)
    t_gen = time.perf_counter() - t_gen
    es_bytes = int(sum(s.size for s in streams))
    n_i = S * ((P + 11) // 12)
    n_p = S * P - n_i

    # (the timed decoder is created before the gate's: a context's streams get their hardware queues at creation, and
    # the first context of a process is dealt the pipes in the order efx_create asks for)
    dec = job.make_decoder(S, P, 2, es_bytes + 64 * S)
    got = gate_against_golden(job.make_decoder, streams, ids, golden, P, f"{workload}, before timing")

    dec.upload(streams, 0)  # bitstreams resident in HBM from here on
    dec.set_timing(True)
    # The launch structure is pinned for the timed steps (efx_set_option): ONE reconstruction group per call -- what back-to-back
    # calls run as anyway, except that the FIRST call after a synchronisation finds the GPU idle and could be split -- so all K
    # calls are the same k_index ... k_recon launches and the per-launch figures are exact.  The parse kernel's residency cap
    # stays automatic: the first call's parser, with nothing to run beside, may have the whole chip (0.84 instead of 1.38 ms
    # of pipeline fill: 2 % of a 20-step region); `launch_structure.parse_cap` says so.
    pinned = hasattr(dec, "set_option") and not args.no_overlap
    if pinned:
        import espflix_amd as efx
        dec.set_option(efx.OPT_GROUPS, 1)
        dec.set_option(efx.OPT_PARSE_CAP, 1 if args.pin_parse_cap else 0)
    elapsed = timed_region(job, dec, steps, warmup, overlap=not args.no_overlap)
    t = dec.timing()
    stage_ms = np.array([t.index_ms, t.parse_ms, t.recon_ms])
    timed_calls = t.timed_calls
    groups = max(1, int(getattr(t, "groups", 1)))  # reconstruction groups of a call
    halves = max(1, int(getattr(t, "parse_halves", groups)))  # parse halves: one k_index ... k_parse sequence each
    # reconstruction kernel launches per call: one k_recon per group and picture index, or one k_recon_all per group
    recon_launches = int(getattr(t, "recon_launches", 0)) or P * groups
    mixed = int(getattr(t, "mixed", 0))
    if mixed:
        log("warning: the timed calls did not all run with the same launch structure: per-launch figures are approximate")

    # the double buffer now holds pictures P-2 and P-1 of every stream: compare them with the reference again
    assert t.pictures == S * P, (t.pictures, S * P)
    bad = [i for i in range(S) if dec.stream_status(i) != 0]
    assert not bad, f"streams with non-zero status: {bad[:8]}"
    hashes = dec.frame_hashes()
    for p in (P - 2, P - 1):
        if p >= 0 and not np.array_equal(hashes[:, dec.picture_slot(p)], got[:, p]):
            raise SystemExit(f"parity gate ({workload}, after timing): picture {p} differs from the reference decoder")
    n_coefs = t.coefficients

    # whole-job check: all-gather the per-stream chain hashes, rank 0 compares with the golden table
    chains = edist.gather_u64(chain_hashes(got), job.dist, job.device, job.world)
    all_ids = edist.gather_u64(np.asarray(ids, dtype=np.uint64), job.dist, job.device, job.world)
    totals = edist.sum_over_ranks([S * P, es_bytes, int(n_coefs)], job.dist, job.device)
    if job.rank == 0:
        want = chain_hashes(golden[all_ids.astype(np.int64) % golden.shape[0]])
        if not np.array_equal(chains, want):
            raise SystemExit(f"parity gate ({workload}): gathered chain hashes differ from the reference on {int((chains != want).sum())} streams")

    # sustained: the driver's K (20 steps = 30 ms) pays the pipeline's fill and drain; the same region over >= 1000 steps
    # is what a service that decodes batch after batch sees.  Gated like the timed region.
    sustained = None
    if sustained_steps and not args.timed_only:
        s_elapsed = timed_region(job, dec, sustained_steps, 0, overlap=not args.no_overlap)
        hashes = dec.frame_hashes()
        for p in (P - 2, P - 1):
            if p >= 0 and not np.array_equal(hashes[:, dec.picture_slot(p)], got[:, p]):
                raise SystemExit(f"parity gate ({workload}, sustained leg): picture {p} differs from the reference decoder")
        sustained = {"steps": sustained_steps, "frames_per_s": totals[0] * sustained_steps / s_elapsed,
                     "ms_per_step": s_elapsed / sustained_steps * 1e3,
                     "what": "the timed region again with this many back-to-back steps (barrier + synchronize on both sides, max "
                             "over ranks), pictures P-2 and P-1 compared with the reference decoder afterwards"}

    # ingest included: every step uploads the batch from host memory again (staging copy, H2D on the copy stream
    # into the bitstream buffer the running decode does not read) and decodes it; upload n+1 overlaps decode n
    ingest = None
    if want_serial and not args.timed_only:
        isteps = max(4, min(steps, 20))

        def ingest_leg(prep, what, in_place=False):
            """in_place: efx_upload_streams_inplace, each upload gated on efx_upload_done() of the one before (the arena's bytes
            belong to the transfer until then); the arena is NOT written again between steps -- a receiver fills it by DMA, and
            a 46 MB host copy per step would time the host's memcpy, not the ingest path."""

            def one_step():
                if in_place:
                    while not dec.upload_done():
                        pass
                    dec.upload_prepared(prep, 0, in_place=True)
                else:
                    dec.upload_prepared(prep, 0)
                dec.decode(sync=False)

            for _ in range(2):
                one_step()
            dec.sync()
            job.barrier()
            t0 = time.perf_counter()
            for _ in range(isteps):
                one_step()
            dec.sync()
            job.barrier()
            dt = edist.max_over_ranks(time.perf_counter() - t0, job.dist, job.device)
            hashes = dec.frame_hashes()
            for p in (P - 2, P - 1):
                if p >= 0 and not np.array_equal(hashes[:, dec.picture_slot(p)], got[:, p]):
                    raise SystemExit(f"parity gate ({workload}, ingest leg, {what}): picture {p} differs from the reference decoder")
            return {"pcie_inclusive_frames_per_s": totals[0] * isteps / dt, "ms_per_step": dt / isteps * 1e3}

        # `pcie_inclusive_frames_per_s` is the STAGED path -- pageable buffers copied into the library's pinned staging memory,
        # then H2D -- as in rounds 1-4 (round-5 ADVICE: the key had moved to the in-place path).  The in-place figure (the batch
        # lies in a page-locked arena of the context in the device layout and every step's H2D reads it where it lies) has its
        # own key, and its leg gates every upload on efx_upload_done()
        ingest = {"steps": isteps}
        ingest.update(ingest_leg(dec.prepare_upload(streams), "staged"))
        ingest["what"] = ("efx_upload_streams (pageable host buffers: threaded staging copy into the library's pinned memory, then H2D on the "
                          "copy stream) + efx_decode per step; three bitstream buffers: the upload of step n+1 runs under the decode of step n")
        if hasattr(dec, "host_arena"):
            arena = dec.host_arena(es_bytes + 32 * S + 4096)
            ingest["in_place_from_page_locked_arena"] = ingest_leg(dec.place_in_arena(arena, streams), "in place", in_place=True)
            ingest["in_place_from_page_locked_arena"]["what"] = (
                "efx_upload_streams_inplace of a batch that lies in page-locked caller memory in the device layout (efx_host_alloc + "
                "efx_stream_layout): ONE H2D per step straight from the caller's bytes, no staging copy; every upload waits for "
                "efx_upload_done() of the one before; the arena's content is laid out once (a receiver fills it by DMA)")

    if pinned:
        dec.set_option(efx.OPT_GROUPS, 0)
        dec.set_option(efx.OPT_PARSE_CAP, 0)
    serial_ms = None
    if want_serial and args.timed_only:
        serial_ms = [float(x) for x in stage_ms]
    elif want_serial:
        # outside the timed region: the same stages one call at a time (no overlap between calls), for
        # the uncontended per-kernel figures quoted next to the timed-region ones
        dec.set_timing(True)
        for _ in range(3):
            dec.decode(sync=True)
        ts = dec.timing()
        serial_ms = [ts.index_ms, ts.parse_ms, ts.recon_ms]
    dec.close()
    with np.errstate(over="ignore"):
        csum = int(np.bitwise_xor.reduce(chains * edist.GOLDEN)) if chains.size else 0
    return {"workload": workload, "S": S, "P": P, "es_bytes": es_bytes, "n_i": n_i, "n_p": n_p, "elapsed": elapsed,
            "stage_ms": stage_ms, "serial_ms": serial_ms, "timed_calls": timed_calls, "groups": groups, "halves": halves, "n_coefs": int(n_coefs),
            "recon_launches": recon_launches, "mixed": mixed, "pinned": bool(pinned),
            "gen_seconds": t_gen, "batch0": batches[0][1], "ingest": ingest, "sustained": sustained, "job_pictures": totals[0], "job_es_bytes": totals[1],
            "checksum": csum, "streams_checked": int(chains.size), "first_id": int(ids[0]), "last_id": int(ids[-1])}


def run_clip(job, args, clip, S, steps):
    """The reference's own embedded clip (src/vmedia.h / src/splash.h, dumped to tests/golden/*.ts),
    replicated S times as transport-stream input: real ffmpeg output, 5 slices per picture.  Gated
    against the reference's per-picture hashes (tests/golden/golden.json), then timed."""
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        g = json.load(f)["clips"][clip]
    want = np.array([int(h, 16) for h in g["hashes"]], dtype=np.uint64)
    P = want.size
    ts = np.fromfile(os.path.join(GOLDEN_DIR, clip + ".ts"), dtype=np.uint8)
    streams = [ts] * S
    dec = job.make_decoder(S, P, P + 1, (ts.size + 64) * S)
    dec.upload(streams, 1)
    dec.decode()
    hashes = dec.frame_hashes()
    for p in range(P):
        if not (hashes[:, dec.picture_slot(p)] == want[p]).all():
            raise SystemExit(f"parity gate (clip {clip}): picture {p} differs from the reference decoder")
    if any(dec.picture_count(i) != P or dec.stream_status(i) for i in range(S)):
        raise SystemExit(f"parity gate (clip {clip}): picture count / status")
    dec.close()
    dec = job.make_decoder(S, P, 2, (ts.size + 64) * S)
    dec.upload(streams, 1)
    dec.set_timing(True)
    elapsed = timed_region(job, dec, steps, 1, overlap=not args.no_overlap)
    t = dec.timing()
    h2 = dec.frame_hashes()
    if not (h2[:, dec.picture_slot(P - 1)] == want[P - 1]).all():
        raise SystemExit(f"parity gate (clip {clip}, after timing): last picture differs")
    dec.close()
    return {"what": f"tests/golden/{clip}.ts (the reference's embedded clip: ffmpeg output, {P} pictures, 5 slices per picture) x {S} as "
                    "transport-stream input (k_demux on upload, not timed), every picture gated against the reference decoder",
            "frames_per_s": S * P * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "pictures_per_stream": int(P),
            "mean_ts_bytes_per_picture": ts.size / P,
            "stage_ms": {"k_index(+scan,emit)": t.index_ms, "k_parse": t.parse_ms, f"k_recon x{P}": t.recon_ms}}


def stage_report(r, steps):
    names = ["k_index(+scan,emit)", "k_parse", "k_recon x%d" % r["P"]]
    out = {"frames_per_s": r["job_pictures"] * steps / r["elapsed"], "ms_per_step": r["elapsed"] / steps * 1e3,
           "mean_bytes_per_picture": r["es_bytes"] / (r["S"] * r["P"]),
           "stage_ms": dict(zip(names, [float(x) for x in r["stage_ms"]]))}
    if r["serial_ms"] is not None:
        out["serial_stage_ms"] = dict(zip(names, [float(x) for x in r["serial_ms"]]))
    return out


SOAK_LEGS = ("primary", "fixed_batch_8192", "wide1500k", "vmedia_x1024", "video_out")


def soak(job, args):
    """--soak LEG N: one process, one leg of the default run, N times -- decoder contexts created and destroyed as the full
    run does -- to chase a rare fault down to a leg (profiles/r5_fault_hunt.md; with EFX_GUARD=1 every device buffer ends on
    an unmapped page, so an over-read faults at once instead of once in forty runs).  Every repetition is gated against the
    reference decoder's goldens like the real leg.  No JSON line: progress goes to stderr."""
    from espflix_amd import dist as edist
    leg, n = args.soak[0], int(args.soak[1])
    if leg not in SOAK_LEGS:
        raise SystemExit(f"--soak: leg must be one of {SOAK_LEGS}")
    S = args.streams
    threads = max(1, usable_cores() // max(1, job.local_world))
    t0 = time.perf_counter()
    for i in range(n):
        if leg == "primary":  # gate, timed region, ingest, one call at a time
            run_workload(job, args, "gop12", S, np.arange(S), threads, 4, 1, sustained_steps=8)
        elif leg == "fixed_batch_8192":
            lo, hi = edist.shard_fixed(job.rank, job.world, args.fixed_batch)
            run_workload(job, args, "gop12", hi - lo, np.arange(lo, hi), threads, 2, 1, want_serial=False)
        elif leg == "wide1500k":
            run_workload(job, args, "wide1500k", S, np.arange(S) % WORKLOADS["wide1500k"][2], threads, 4, 1)
        elif leg == "vmedia_x1024":
            run_clip(job, args, "vmedia", S, 2)
        elif leg == "video_out":
            args.no_cpu_baseline = True
            run_video_out(job, args, S)
        log(f"soak {leg}: {i + 1}/{n} clean, {time.perf_counter() - t0:.1f} s")
    log(f"SOAK_OK {leg} {n}")


def run(job, args):
    from espflix_amd import dist as edist
    rank, world = job.rank, job.world
    S = args.streams
    threads = max(1, usable_cores() // max(1, job.local_world))
    first, _ = edist.shard(rank, world, S)
    ids = np.arange(first, first + S)
    if world > 1 and rank == 0:
        log(f"{'RCCL' if job.device == 'cuda' else 'gloo'} saw {world} ranks")

    r = run_workload(job, args, "gop12", S, ids, threads, args.steps, args.warmup, sustained_steps=args.sustained_steps)
    P = r["P"]

    # SURVEY 8d config 5 as written: the FIXED 8192-stream batch, stream k on rank floor(k*N/8192)
    log("leg: primary done; fixed batch")
    fixed = None
    if not args.no_fixed_batch:
        if S * world == args.fixed_batch:
            fixed = {"same_as_primary": True}
        else:
            lo, hi = edist.shard_fixed(rank, world, args.fixed_batch)
            fsteps = max(2, min(args.steps, 10))
            fr = run_workload(job, args, "gop12", hi - lo, np.arange(lo, hi), threads, fsteps, 2, want_serial=False)
            fixed = {"steps": fsteps}
            fixed.update(stage_report(fr, fsteps))
            fixed["streams_per_gpu"] = hi - lo
        fixed["streams_total"] = args.fixed_batch
        fixed["partition"] = f"stream k -> rank floor(k*{world}/{args.fixed_batch})"
        fixed["scaling"] = "strong"

    log("leg: other workloads")
    others = None
    if world == 1 and not args.no_other_workloads:
        others = {}
        ow = run_workload(job, args, "wide1500k", S, np.arange(S) % WORKLOADS["wide1500k"][2], threads, min(args.steps, 10), 2)
        others["wide_slices_1500k"] = {"what": f"{S} streams x GOP(12), 5 slices per picture spanning 2-3 macroblock rows, quantiser "
                                               "chosen per stream for ~6.25 kB per picture (the service's 1.5 Mbit/s profile, reference "
                                               "indexer/indexer.cpp:306-309), every picture gated against the reference decoder",
                                       **stage_report(ow, min(args.steps, 10))}
        # The parse-parallelism lever of this shape (round-5 verdict item 4): 5 slices x 12 pictures x 1024 streams are 61 k parse
        # lanes -- 960 waves, less than one per SIMD -- so the 12-picture call is bound by the longest lanes' token chains; the
        # same streams at 36 and 72 pictures per call (the streaming adapter cuts windows of up to 30)
        # give the parser three and six times the lanes.  Every picture gated against the reference's hashes of these streams.
        curve = {12: {"frames_per_s": others["wide_slices_1500k"]["frames_per_s"], "ms_per_step": others["wide_slices_1500k"]["ms_per_step"],
                      "serial_stage_ms": others["wide_slices_1500k"].get("serial_stage_ms")}}
        for pics in (36, 72):
            wsteps = max(4, min(args.steps, 10))  # (as many CALLS as the 12-picture leg: a region of two calls is all fill and drain)
            o2 = run_workload(job, args, f"wide1500k_p{pics}", S, np.arange(S) % 1024, threads, wsteps, 1)
            rep = stage_report(o2, wsteps)
            curve[pics] = {"frames_per_s": rep["frames_per_s"], "ms_per_step": rep["ms_per_step"], "serial_stage_ms": rep.get("serial_stage_ms"),
                           "steps": wsteps}
        others["wide_slices_1500k"]["pictures_per_call"] = {
            "what": "the same 5-slice 1.5 Mbit/s streams at 12 / 36 / 72 pictures per stream and efx_decode call (one, three, six GOPs): "
                    "frames/s of back-to-back calls and the stages one call at a time",
            **{str(k): v for k, v in curve.items()}}
        others["vmedia_x%d" % S] = run_clip(job, args, "vmedia", S, max(2, min(args.steps, 5)))

    log("leg: video out")
    video_out = None
    if world == 1 and job.device == "cuda" and not args.no_video_out:
        video_out = run_video_out(job, args, S)

    cpu = cpu_o3 = None
    if rank == 0 and not args.no_cpu_baseline:
        log("leg: cpu baseline")
        cpu = cpu_baseline(r["batch0"], S, P, 1024, args.cpu_baseline_seconds)
        if job.device == "cuda":
            cpu_o3 = cpu_baseline_o3(r["batch0"], S, P, 1024, args.cpu_baseline_seconds)
    job.barrier()

    if rank != 0:
        return None
    steps = args.steps
    value = r["job_pictures"] * steps / r["elapsed"]
    stage_ms = r["stage_ms"]
    names = ["k_index(+scan,emit)", "k_parse", "k_recon x%d" % P]
    alg = algorithmic_bytes(r["es_bytes"], r["n_i"], r["n_p"])
    # The roofline object is k_recon's: it is the kernel that moves SURVEY 8d's algorithmic bytes (the
    # frames); k_parse, which only reads the bitstream and writes 4 B per coefficient + 16 B per
    # macroblock, is reported next to it.
    # libefx runs a call as G groups of streams; a group is reconstructed by ONE launch (k_recon_all: every picture index, the
    # pictures of a stream ordered by a per-stream counter) or by one k_recon launch per picture index (EFX_OPT_RECON_MODE 0)
    G = r["groups"]
    L = r.get("recon_launches") or P * G
    recon_kernel = "k_recon_all" if L == G else "k_recon"
    names[2] = f"{recon_kernel} x{L // G}" if L == G else names[2]
    alg_launch = alg / L
    dur_s = stage_ms[2] / 1e3 / L
    achieved = alg_launch / dur_s / 1e9
    parse_bytes = r["es_bytes"] + 4 * r["n_coefs"] + 16 * S * P * 264
    traffic, traffic_src = pmc_traffic("efx::" + recon_kernel, S, P)
    ptraffic, _ = pmc_traffic("efx::k_parse", S, P)
    serial_ms = r["serial_ms"]
    # the issue floor of the kernels of a step, ONE convention (issue_model): dynamic vector instructions x the kernel's
    # mix-weighted cycles per instruction / (1024 SIMDs x 2.4 GHz)
    im = issue_model()
    issue = None
    bound = "hbm"
    kr, kp = im.get("k_recon"), im.get("k_parse")
    if kr and kr.get("dynamic_valu_per_launch") and recon_kernel == "k_recon" and S // G == 1024:
        floor_launch_s = kr["dynamic_valu_per_launch"] * kr["cycles_per_valu"] / (SIMDS * CLOCK_HZ)
        parse_floor_s = (kp["dynamic_valu_per_launch"] * kp["cycles_per_valu"] / (SIMDS * CLOCK_HZ)) if kp and kp.get("dynamic_valu_per_launch") else None
        step_floor_s = floor_launch_s * L + (parse_floor_s or 0.0)
        issue = {"unit": "SIMD issue cycles", "convention": "full-rate opcodes 2.2 cycles per wave64 instruction and SIMD, half-rate 4.2 "
                 "(profiles/r6_valu_rates.md); floor = SQ_INSTS_VALU x mix-weighted cycles per instruction / (1024 SIMDs x 2.4 GHz)",
                 "cycles_per_instruction": kr["cycles_per_valu"], "vector_instructions_per_launch": kr["dynamic_valu_per_launch"],
                 "floor_us_per_launch": floor_launch_s * 1e6, "frac": floor_launch_s / dur_s,
                 "serial_frac": floor_launch_s * L / (serial_ms[2] / 1e3),
                 "k_parse_floor_us": parse_floor_s * 1e6 if parse_floor_s else None,
                 "whole_step_floor_ms": step_floor_s * 1e3, "whole_step_frac": step_floor_s / (r["elapsed"] / steps),
                 "source": "profiles/r6_issue_floor.json (tools/issue_floor.py)"}
        # what the numbers say: the larger of the two fractions names the nearer roof
        bound = "valu_issue" if issue["frac"] > achieved / HBM_PEAK_GBS else "hbm"
    out = {
        "metric": "MPEG-1 352x192 frames/s", "value": value, "unit": "frames/s", "n_gpus": world,
        "steps": steps, "warmup": args.warmup, "ms_per_step": r["elapsed"] / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
        "config": {"workload": f"{S} streams/GPU x GOP(12) I+11P, 352x192 MPEG-1 ES resident in HBM, ids rank*{S}.. per rank "
                               f"(BASELINE configs[2]; at 8 GPUs = configs[4], stream k on rank floor(k*8/8192))",
                   "streams_per_gpu": S, "streams_total": S * world, "pictures_per_stream": P, "es_bytes_per_gpu": r["es_bytes"],
                   "mean_bytes_per_picture": r["es_bytes"] / (S * P), "parallelism": f"stream-partition x{world}",
                   "coefficients_per_gpu": r["n_coefs"], "ring_depth": 2},
        "roofline": {"bound": bound, "bound_note": "achieved / peak / frac are SURVEY 8d's algorithmic bytes against the HBM roofline, as the "
                     "contract asks; the resource this kernel is nearest to is vector-instruction ISSUE (`issue`): its arithmetic at the "
                     "measured opcode rates (profiles/r6_valu_rates.md) fills issue.frac of the chip's SIMD cycles in the timed region and "
                     "issue.serial_frac one call at a time, against %.2f of the HBM roofline -- the rest of a launch is the waves' dependent "
                     "round trips at 12-14 resident waves per CU and the launch's fill and drain (profiles/r6_recon_vmem.md: neither fewer "
                     "vector-memory instructions nor fewer vector instructions shorten it any more)" % (achieved / HBM_PEAK_GBS),
                     "kernel": names[2], "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "issue": issue,
                     "algorithmic_bytes_per_launch": alg_launch, "avg_launch_ms": dur_s * 1e3, "launches_per_step": L,
                     "streams_per_launch": S // G, "pictures_per_launch": P * G // L,
                     "launch_structure": {"pinned": r.get("pinned", False), "mixed": r.get("mixed", 0), "groups": G, "parse_halves": r["halves"],
                                          "parse_cap": "every call" if args.pin_parse_cap else "automatic: every call that finds reconstruction queued (all but the first)"},
                     "whole_step_achieved_GBs": alg / (r["elapsed"] / steps) / 1e9,
                     "whole_step_frac": alg / (r["elapsed"] / steps) / 1e9 / HBM_PEAK_GBS,
                     "stage_ms": dict(zip(names, [float(x) for x in stage_ms])),
                     "timed_calls_averaged": r["timed_calls"],
                     "note": "stage_ms / avg_launch_ms are means over the timed steps, where the parse half of "
                             "later steps shares the GPU with k_recon; serial_* = the same stages one call at a "
                             "time, measured after the timed region",
                     "serial_stage_ms": dict(zip(names, [float(x) for x in serial_ms])),
                     "serial_frac": alg / (serial_ms[2] / 1e3) / 1e9 / HBM_PEAK_GBS,
                     "k_parse": {"algorithmic_bytes_per_launch": parse_bytes / r["halves"], "avg_launch_ms": float(stage_ms[1]) / r["halves"],
                                 "launches_per_step": r["halves"],
                                 "achieved": parse_bytes / (stage_ms[1] / 1e3) / 1e9,
                                 "serial_launch_ms": float(serial_ms[1]) / r["halves"], "traffic": ptraffic,
                                 "bound": "serial token chain of the heaviest lane of each wave (270 ns per trip alone, 326 beside k_recon), not bandwidth"}},
        "parity_gate": {"reference": "tests/golden/bench_gop12.u64 (unmodified reference decoder, tests/golden/make_bench_golden.py)",
                        "streams_checked": r["streams_checked"], "pictures_checked_per_stream": P,
                        "when": "every picture before the timed region; pictures P-2, P-1 again after it; chain hashes of all "
                                "ranks gathered and compared on rank 0", "passed": True},
        "checksum_of_checksums": f"{r['checksum']:016x}",
        "gen_seconds": r["gen_seconds"],
        "sustained": r["sustained"],
        "ingest": r["ingest"],
        "fixed_batch_8192": fixed,
        "other_workloads": others,
        "video_out": video_out,
        "cpu_baseline": cpu,
        "cpu_baseline_o3": cpu_o3,
    }
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: run the same command line under
    torch.distributed.run, one rank per GPU, and pass its output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU")
    ap.add_argument("--fixed-batch", type=int, default=FIXED_BATCH, help="streams of the fixed (strong-scaling) batch")
    ap.add_argument("--no-fixed-batch", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-video-out", action="store_true", help="skip the composite / PDM / demux / SBC legs (N = 1 only anyway)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=5.0, help="wall-time target of the CPU baseline leg")
    ap.add_argument("--sustained-steps", type=int, default=1000, help="steps of the sustained leg beside the K timed ones (0: skip)")
    ap.add_argument("--no-overlap", action="store_true", help="synchronise after every step (no cross-step pipelining)")
    ap.add_argument("--pin-parse-cap", action="store_true", help="timed region: cap the parse kernel's residency on every call, the first after a "
                                                                 "synchronisation included (every k_parse launch then has the same grid)")
    ap.add_argument("--soak", nargs=2, metavar=("LEG", "N"), help=f"repeat one leg N times in this process (legs: {', '.join(SOAK_LEGS)})")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling aid: skip the ingest and one-call-at-a-time legs so that (nearly) every kernel launch of the "
                         "process belongs to the timed region (serial_* fields then repeat the timed-region figures)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import espflix_amd as efx
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: espflix_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    # one process per GPU: this rank's threads (and the pinned staging memory its decoder contexts will first touch) go to the
    # NUMA node its device hangs off -- at 8 ranks per node the ingest leg is a host-memory problem (espflix_amd/dist.py)
    numa = None
    if os.environ.get("EFX_NUMA", "1") != "0":
        try:
            from espflix_amd import dist as edist
            pr = torch.cuda.get_device_properties(local_rank)
            bus = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            numa = dict(edist.bind_rank_to_device_node(bus), pci=bus)
            log(f"NUMA: device {local_rank} at {bus}: {numa}")
        except Exception as e:  # (placement is for speed only)
            log(f"NUMA: not bound ({e})")
    dist = None
    if world > 1 or os.environ.get("EFX_BENCH_FORCE_DIST"):  # (the variable: the RCCL code path with a single rank, for testing)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def make_decoder(max_streams, max_pictures, ring_depth, max_stream_bytes):
        return efx.Decoder(max_streams=max_streams, max_pictures=max_pictures, ring_depth=ring_depth, device=local_rank,
                           max_stream_bytes=max_stream_bytes)

    job = Job(rank, world, dist, "cuda", make_decoder, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.soak:
        soak(job, args)
        return
    out = run(job, args)
    if rank == 0:
        out["config"]["numa"] = numa
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
