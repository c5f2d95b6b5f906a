#!/bin/bash
# development aid: SQ / TCP / TA / TCC PMC passes of a short serial bench run, per-kernel means printed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-fixed-batch --no-other-workloads"
run() { n=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pm_${tag}_$n -o p -- $B > /dev/null 2>&1; }
run sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM
run tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run g GRBM_GUI_ACTIVE
python - <<P
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pm_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('efx::k_recon') or k.startswith('efx::k_cl'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(cs.items())}, '(millions)')
P
