#!/bin/bash
# round 5, run 18: counters of k_sbc_par_mono (mono clean, 1024 streams x 375 frames)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5w
k=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  k=$((k+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/r5w/pmc_$k -o p -- python tools/exp/r5_sbc.py mono_clean > gpurun_out/r5w/pmc_$k.log 2>&1
done
python - <<'PY' | tee gpurun_out/r5w/sbc_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r5w/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "k_sbc" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"   {c:28s} per launch {sum(v) / max(1, len(v)) if False else 0:0.0f}" if False else f"   {c:28s} n={len(v):4d} mean {sum(v)/len(v):16.1f}")
PY
rm -rf gpurun_out/r5w/pmc_?
