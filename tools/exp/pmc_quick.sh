#!/bin/bash
# development aid: three PMC passes (SQ / WRITE_SIZE / FETCH_SIZE) of a short serial bench run, per-kernel means printed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-fixed-batch --no-other-workloads"
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pq_${tag}_SQ -o p -- $B > /dev/null 2>&1
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pq_${tag}_$c -o p -- $B > /dev/null 2>&1
done
python - <<P
import csv, os, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pq_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('efx::'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, 'n=%d' % len(next(iter(cs.values()))))
P
