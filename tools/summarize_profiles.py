#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py <tag> --stats DIR/PREFIX ... --pmc DIR ...

Writes profiles/<tag>_kernel_stats_<name>.csv (verbatim rocprofv3 --stats table) and
profiles/<tag>_pmc_summary.json: per kernel the mean of every counter over its dispatches.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B; on gfx950
FETCH_SIZE counts 128-byte requests as 64 B (MI355X_MICROARCH.md, HBM section) -- calibrated
here on k_frame_hash, which reads every ring frame exactly once -- so hbm_read_bytes = 2 x
FETCH_SIZE x 1024; WRITE_SIZE is taken as reported (partial-line writes are counted as 32 B
requests, which over-counts kernels that store narrow per-lane words).
"""
import argparse, collections, csv, json, os, shutil, sys

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--stats", nargs="*", default=[])
ap.add_argument("--pmc", nargs="*", default=[])
ap.add_argument("--video-pmc", nargs="*", default=[], help="PMC passes of tools/bench_video.py (k_composite NTSC / PAL, k_pdm ...)")
ap.add_argument("--extra", nargs="*", default=[], metavar="NAME=DIR[,DIR...]",
                help="further PMC passes summarised the same way into their own section NAME (round 5: the uncapped parse "
                     "schedule, the TA / TCP / LDS counters of k_recon at two LDS footprints)")
ap.add_argument("--out", default=None, help="directory to write into (default: profiles/ of the repository)")
ap.add_argument("--note", default="")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = a.out or os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

for s in a.stats:
    name = os.path.basename(s)
    shutil.copy(s + "_kernel_stats.csv", os.path.join(out, f"{a.tag}_kernel_stats_{name}.csv"))

def summarise(dirs):
    summary = collections.defaultdict(dict)
    for d in dirs:
        rows = collections.defaultdict(list)
        if not os.path.isdir(d):
            continue
        for f in os.listdir(d):
            if not f.endswith("counter_collection.csv"):
                continue
            for r in csv.DictReader(open(os.path.join(d, f))):
                k = r["Kernel_Name"].split("(")[0]
                if k.startswith("efx::"):
                    rows[k].append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
        for k, rs in rows.items():
            grid = collections.Counter(g for g, _, _ in rs).most_common(1)[0][0]
            summary[k]["grid_size"] = grid
            cs = collections.defaultdict(list)
            for g, c, v in rs:
                if g == grid:
                    cs[c].append(v)
            for c, v in cs.items():
                summary[k][c] = sum(v) / len(v)
                summary[k]["dispatches_" + c] = len(v)
    for k, cs in summary.items():
        if "FETCH_SIZE" in cs:
            cs["hbm_read_bytes"] = 2.0 * cs["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in cs:
            cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0
        if "hbm_read_bytes" in cs and "hbm_write_bytes" in cs:
            cs["hbm_traffic_bytes"] = cs["hbm_read_bytes"] + cs["hbm_write_bytes"]
    return summary


extra = {}
for spec in a.extra:
    name, dirs = spec.split("=", 1)
    extra[name] = summarise(dirs.split(","))

summary = collections.defaultdict(dict)
for d in a.pmc:
    # per kernel only the dispatches of its most frequent grid size: libefx runs the first call of a context as one
    # group of streams and later ones as groups of ~512 (efx_decode_from), and a mean over both would be neither
    rows = collections.defaultdict(list)
    for f in os.listdir(d):
        if not f.endswith("counter_collection.csv"):
            continue
        for r in csv.DictReader(open(os.path.join(d, f))):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("efx::"):
                rows[k].append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    for k, rs in rows.items():
        grid = collections.Counter(g for g, _, _ in rs).most_common(1)[0][0]
        summary[k]["grid_size"] = grid
        cs = collections.defaultdict(list)
        for g, c, v in rs:
            if g == grid:
                cs[c].append(v)
        for c, v in cs.items():
            summary[k][c] = sum(v) / len(v)
            summary[k]["dispatches_" + c] = len(v)
for k, cs in summary.items():
    if "FETCH_SIZE" in cs:
        cs["hbm_read_bytes"] = 2.0 * cs["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in cs:
        cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes" in cs and "hbm_write_bytes" in cs:
        cs["hbm_traffic_bytes"] = cs["hbm_read_bytes"] + cs["hbm_write_bytes"]
# video-out kernels (tools/bench_video.py): one entry per (kernel, grid size) -- k_composite runs with one grid for NTSC
# (17 blocks of 16 lines per stream) and a larger one for PAL (20)
video = collections.defaultdict(dict)
for d in a.video_pmc:
    rows = collections.defaultdict(list)
    for f in os.listdir(d):
        if not f.endswith("counter_collection.csv"):
            continue
        for r in csv.DictReader(open(os.path.join(d, f))):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("efx::"):
                rows[(k, int(r["Grid_Size"]))].append((r["Counter_Name"], float(r["Counter_Value"])))
    comp = sorted(g for (k, g) in rows if k == "efx::k_composite")
    big = [g for g in comp if len(rows[("efx::k_composite", g)]) >= 10]  # (the one-stream gate launches are not the timed ones)
    for (k, g), rs in rows.items():
        name = k
        if k == "efx::k_composite":
            if g not in big[-2:]:
                continue
            name = k + (":ntsc" if g == big[-2] else ":pal")
        elif len(rs) < 2 and k != "efx::k_pdm":
            continue
        cs = collections.defaultdict(list)
        for c, v in rs:
            cs[c].append(v)
        video[name]["grid_size"] = g
        for c, v in cs.items():
            video[name][c] = sum(v) / len(v)
            video[name]["dispatches_" + c] = len(v)
for k, cs in video.items():
    if "FETCH_SIZE" in cs:
        cs["hbm_read_bytes"] = 2.0 * cs["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in cs:
        cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0
    if "hbm_read_bytes" in cs and "hbm_write_bytes" in cs:
        cs["hbm_traffic_bytes"] = cs["hbm_read_bytes"] + cs["hbm_write_bytes"]

if summary or video:
    sys.path.insert(0, root)
    import bench  # (kernel_sources_digest: what the counters were measured on; bench.py warns when the sources move on)
    doc = {"note": a.note, "kernel_sources_digest": bench.kernel_sources_digest(), "kernel_source_files": bench.kernel_source_digests(),
           "kernels": summary, "video_kernels": video}
    doc.update(extra)
    json.dump(doc, open(os.path.join(out, f"{a.tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in doc.items() if k != "note"}, indent=1, sort_keys=True))
