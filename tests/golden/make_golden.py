#!/usr/bin/env python3
"""Generates tests/golden/{splash.ts,vmedia.ts,golden.json} by RUNNING THE UNMODIFIED REFERENCE
(oracle/_ref, built by `make ref` from /root/reference).  Run in the build container only:

    make ref gen oracle && python tests/golden/make_golden.py

Everything in golden.json is an output of the reference itself:
  clips      per-frame FNV-1a-64 + pts of every frame push_video() received for the two clips
             embedded in the reference (src/splash.h, src/vmedia.h), with and without the final
             flush_picture(1)
  synthetic  the same for generator streams (TS-wrapped) of several flavours
  handmade   the same for streams written bit by bit in tests/common.py (macroblock_stuffing / macroblock_escape runs)
  display    FNV of video_isr() fields with _hscroll slides and the composite() overlay / progress bar
  index      FNV of the indexer's video.idx for synthetic titles and the clips
  sbc        FNV of sbc_decoder() PCM for synthetic frame configurations and the clips' PID 0x102 audio
  composite  FNV of video_isr() fields (NTSC and PAL, 3 fields) for LCG / random / decoded frames
  pdm        FNV of write_pcm_16() output incl. silence and beep calls
  tables     zig_zag, scale_dct_q, _color_tab, video geometry
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
import oracle
from espflix_amd import gen

assert oracle.have_ref(), "build oracle/_ref first (make ref)"
out = {"clips": {}, "synthetic": {}, "handmade": {}, "composite": {}, "display": {}, "pdm": {}, "sbc": {}, "index": {}, "tables": {}}

for clip in ("splash", "vmedia"):
    subprocess.run([os.path.join(oracle.REF_DIR, "efx_ref_decode"), "fixture", "@" + clip,
                    os.path.join(HERE, clip + ".ts")], check=True)
    h, pts, _ = oracle.ref_decode("@" + clip, flush_last=True)
    h2, _, _ = oracle.ref_decode("@" + clip, flush_last=False)
    assert len(h2) == len(h) - 1 and (h2 == h[:-1]).all()
    out["clips"][clip] = {"hashes": [f"{int(x):016x}" for x in h], "pts": [int(x) for x in pts],
                          "pushed_without_flush": len(h2), "chain": f"{oracle.chain_hash(h2):016x}"}

for flags in common.SYN_FLAGS:
    b = gen.Batch(0, 8, 12, 12, flags)
    for k in common.SYN_IDS:
        h, pts, _ = oracle.ref_decode(b.ts(k), flush_last=True)
        out["synthetic"][f"{flags}:{k}"] = {"hashes": [f"{int(x):016x}" for x in h], "pts": [int(x) for x in pts],
                                            "es_fnv": f"{common.fnv_bytes(b.es(k)):016x}"}

# macroblock_stuffing in every number, address escapes, stuffing behind an escape (player.cpp:1267-1275)
es = common.stuffing_es()
h, pts, _ = oracle.ref_decode(np.frombuffer(common.one_pes_per_picture(es), dtype=np.uint8), flush_last=True)
out["handmade"]["stuffing"] = {"hashes": [f"{int(x):016x}" for x in h], "pts": [int(x) for x in pts],
                               "es_fnv": f"{common.fnv_bytes(np.frombuffer(es, dtype=np.uint8)):016x}"}

_, _, frames = oracle.ref_decode(gen.Batch(0, 1, 12, 12, 0).ts(0), flush_last=True, want_frames=True)
inputs = {"lcg": common.lcg_frames(), "random": common.random_frames(7), "decoded": np.concatenate([frames[10], frames[11]])}
for name, fr in inputs.items():
    for ntsc in (True, False):
        f = oracle.ref_video_field(fr, ntsc, 3)
        out["composite"][f"{name}:{'ntsc' if ntsc else 'pal'}"] = [f"{common.fnv_bytes(f[i]):016x}" for i in range(3)]

# display state: two-frame slide (_hscroll) and overlay / progress bar, on random frames <= 248
disp = np.minimum(common.random_frames(77), 248)
for name, front, hs, ov_seed, blend, progress in common.DISPLAY_CASES:
    n = len(hs) if hs is not None else 6
    ov = common.overlay_bytes(ov_seed) if ov_seed is not None else (np.zeros(1280, np.uint8) if blend else None)
    for ntsc in (True, False):
        f = oracle.ref_video_field_ex(disp, ntsc, n, front, hs, ov, blend, progress)
        out["display"][f"{name}:{'ntsc' if ntsc else 'pal'}"] = [f"{common.fnv_bytes(f[i]):016x}" for i in range(n)]

pcm = common.pdm_pcm(0, 40)
out["pdm"]["sine220_silence7_beep3"] = f"{common.fnv_bytes(oracle.ref_pdm(pcm, silence_every=7, beep_at=3)):016x}"
out["pdm"]["sine220"] = f"{common.fnv_bytes(oracle.ref_pdm(pcm)):016x}"

# trick-play index: video.idx as the reference indexer writes it (struct padding masked)
clip_ts = {c: np.fromfile(os.path.join(HERE, c + ".ts"), dtype=np.uint8) for c in ("splash", "vmedia")}
for name, streams in common.index_titles() + [("clips", [clip_ts["vmedia"], clip_ts["splash"], clip_ts["vmedia"]])]:
    out["index"][name] = f"{oracle.fnv1a64(oracle.idx_masked(oracle.ref_make_idx(streams))):016x}"

# SBC audio: PCM of the reference's sbc_decoder() on synthetic frames and on the clips' own audio
for name, kw, n, probe in common.SBC_CASES:
    fr = common.sbc_frames(common.seed_of(name), n, **kw)
    fb = common.sbc_frame_bytes(kw["blocks"], 1 if kw["mode"] == 0 else 2, kw["bitpool"])
    pcm_ref, _ = oracle.ref_sbc_decode(fr, fb, probe)
    out["sbc"][name] = f"{common.fnv_bytes(pcm_ref):016x}"
for clip in ("splash", "vmedia"):
    ts = np.fromfile(os.path.join(HERE, clip + ".ts"), dtype=np.uint8)
    es = oracle.ts_audio_es(ts)
    fb = common.CLIP_SBC_FRAME_BYTES[clip]
    pcm_ref, _ = oracle.ref_sbc_decode(es[:es.size // fb * fb], fb, True)
    out["sbc"]["clip:" + clip] = {"frames": int(es.size // fb), "audio_es_fnv": f"{common.fnv_bytes(es):016x}",
                                  "pcm_fnv": f"{common.fnv_bytes(pcm_ref):016x}"}
syn, pro = oracle.ref_sbc_tables()
out["tables"]["sbc_syn_8"] = f"{common.fnv_bytes(syn):016x}"
out["tables"]["sbc_proto_8"] = f"{common.fnv_bytes(pro):016x}"

for ntsc in (True, False):
    params, ctab, dither = oracle.ref_video_params(ntsc)
    out["tables"]["params_" + ("ntsc" if ntsc else "pal")] = [int(x) for x in params]
    out["tables"]["color_tab_" + ("ntsc" if ntsc else "pal")] = f"{common.fnv_bytes(ctab):016x}"
    out["tables"]["dither4x4"] = [int(x) for x in dither]
tb = os.path.join(HERE, "_t.bin")
subprocess.run([os.path.join(oracle.REF_DIR, "efx_ref_decode"), "tables", tb], check=True)
t = np.fromfile(tb, dtype=np.uint8)
os.remove(tb)
out["tables"]["zig_zag"] = [int(x) for x in t[:64]]
out["tables"]["scale_dct_q"] = [int(x) for x in t[64:]]

with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote golden.json:", {k: len(v) for k, v in out.items()})
