#!/bin/bash
# development aid: the primary bench line for several batch sizes (streams per GPU)
for n in "$@"; do
  python bench.py --streams $n --steps 100 --warmup 5 --no-fixed-batch --no-cpu-baseline --no-other-workloads > gpurun_out/bs_$n.json 2> gpurun_out/bs_$n.err || tail -2 gpurun_out/bs_$n.err
  python - <<P
import json
d=json.load(open('gpurun_out/bs_$n.json')); r=d['roofline']
f=lambda m:[round(v,3) for v in m.values()]
print('$n', '%.2f M frames/s'%(d['value']/1e6), 'ms/step %.3f'%d['ms_per_step'], 'serial', f(r['serial_stage_ms']))
P
done
