#!/bin/bash
# development aid: rocprofv3 kernel-trace stats of a short serial bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}; shift
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fixed-batch --no-other-workloads $@"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$tag -o s -- $B > gpurun_out/st_$tag.log 2>&1
f=$(find gpurun_out/st_$tag -name "*kernel_stats.csv" | head -1)
python - <<P
import csv
for r in csv.DictReader(open("$f")):
    if 'efx::' in r['Name']:
        print(r['Name'].split('(')[0], 'calls', r['Calls'], 'avg_us %.1f'%(float(r['AverageNs'])/1e3), 'min %.1f max %.1f'%(float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3), 'total_ms %.2f'%(float(r['TotalDurationNs'])/1e6))
P
