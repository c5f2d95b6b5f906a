#!/bin/bash
# development aid: instruction-fetch counters of k_recon / k_parse (serial mode: counter collection serialises dispatches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --no-overlap"
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmci_$name -o p -- $S > gpurun_out/pmci_$name.log 2>&1; echo "$name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES
run if SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
run sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
python - <<'P'
import csv, collections, os
for name in ("ic", "if", "sq"):
    d = f"gpurun_out/pmci_{name}"
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in os.listdir(d):
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                k = r["Kernel_Name"].split("(")[0]
                if k in ("efx::k_recon", "efx::k_parse"):
                    rows[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in sorted(rows.items()):
        print(name, k, "grid", g, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
P
