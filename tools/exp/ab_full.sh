#!/bin/bash
# development aid: primary line + ingest + the two real-shape workloads for the libefx builds named on the command line
for v in "$@"; do
  export EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  python bench.py --steps 40 --warmup 5 --no-fixed-batch --no-cpu-baseline > gpurun_out/abf_$v.json 2> gpurun_out/abf_$v.err || tail -3 gpurun_out/abf_$v.err
  python - <<P
import json
d=json.load(open('gpurun_out/abf_$v.json')); r=d['roofline']; o=d['other_workloads']
print('$v', 'gop12 %.2fM'%(d['value']/1e6), 'ingest %.2fM'%(d['ingest']['pcie_inclusive_frames_per_s']/1e6), ' '.join('%s %.2fM'%(k,v['frames_per_s']/1e6) for k,v in o.items()))
P
done
