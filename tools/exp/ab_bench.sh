#!/bin/bash
# development aid: the bench's primary line (pipelined + serial stage times) for the libefx builds named on the command line
for v in "$@"; do
  export EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  python bench.py --steps 30 --warmup 5 --no-fixed-batch --no-cpu-baseline --no-other-workloads > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || tail -3 gpurun_out/ab_$v.err
  python - <<P
import json
d=json.load(open('gpurun_out/ab_$v.json')); r=d['roofline']
f=lambda m:[round(v,3) for v in m.values()]
print('$v', 'value %.2fM'%(d['value']/1e6), 'pipelined', f(r['stage_ms']), 'serial', f(r['serial_stage_ms']), 'ingest %.2fM'%(d['ingest']['pcie_inclusive_frames_per_s']/1e6))
P
done
