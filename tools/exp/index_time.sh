#!/bin/bash
# development aid: k_index launch durations (rocprofv3 kernel trace) for libefx_<x>.so builds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
export EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/it_$v -o s -- python tools/exp/time_variants.py $v > gpurun_out/it_$v.log 2>&1
python - <<P
import csv, glob
f = glob.glob('gpurun_out/it_$v/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_index' in r['Name'] or 'k_slice' in r['Name']:
        print('$v', r['Name'].split('(')[0], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
P
done
