#!/bin/bash
# development aid: memory REQUEST counters of k_recon / k_parse (serial mode: counter collection serialises dispatches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --no-overlap"
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcm_$name -o p -- $S > gpurun_out/pmcm_$name.log 2>&1; echo "$name rc=$?"; }
run tcc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
run ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run tcp2 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TOTAL_CACHE_ACCESSES_sum
python - <<'P'
import csv, collections, os
for name in ("tcc", "tcp", "ta", "tcp2"):
    d = f"gpurun_out/pmcm_{name}"
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.isdir(d):
        print(name, "no output"); continue
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    k = r["Kernel_Name"].split("(")[0]
                    if k in ("efx::k_recon", "efx::k_parse", "efx::k_index"):
                        rows[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in sorted(rows.items()):
        print(name, k, "grid", g, {c: f"{sum(v)/len(v):.4g} (n={len(v)})" for c, v in cs.items()})
    if not rows:
        os.system(f"tail -3 gpurun_out/pmcm_{name}.log")
P
