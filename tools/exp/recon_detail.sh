#!/bin/bash
# development aid: k_recon launch durations by picture index (serial run) + one SQ counter pass, for libefx_<x>.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
export EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap --no-fixed-batch --no-other-workloads"
timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rd_$v -o s -- $B > gpurun_out/rd_$v.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/rdp_$v -o p -- $B > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/rdq_$v -o p -- $B > /dev/null 2>&1
python - <<P
import csv, glob, collections
f = glob.glob('gpurun_out/rd_$v/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_recon' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
by = collections.defaultdict(list)
for i, x in enumerate(d): by[i % 12].append(x)
print('$v', 'k_recon us by picture index:', [round(sum(v) / len(v), 1) for k, v in sorted(by.items())], 'mean %.1f' % (sum(d) / len(d)))
for pat in ('rdp', 'rdq'):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/%s_$v/**/*counter_collection.csv' % pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_recon' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('   ', {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(agg.items())}, '(millions per launch)')
P
done
