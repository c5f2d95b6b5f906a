#!/bin/bash
# development aid: the bench's primary line for several values of one environment variable: ab_env.sh VAR v1 v2 ...
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export $var=$v
  python bench.py --steps 200 --warmup 5 --no-fixed-batch --no-cpu-baseline --no-other-workloads > gpurun_out/abe_$v.json 2> gpurun_out/abe_$v.err || tail -3 gpurun_out/abe_$v.err
  python - <<P
import json
d=json.load(open('gpurun_out/abe_$v.json')); r=d['roofline']
f=lambda m:[round(v,3) for v in m.values()]
print('$var=$v', 'value %.2fM'%(d['value']/1e6), 'pipelined', f(r['stage_ms']), 'serial', f(r['serial_stage_ms']), 'ingest %.2fM'%(d['ingest']['pcie_inclusive_frames_per_s']/1e6))
P
done
done
