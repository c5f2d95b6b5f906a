# quick same-box check: GPU parity tests, then the bench without the long legs
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-fixed-batch --no-cpu-baseline > gpurun_out/q.json 2> gpurun_out/q.err || tail -5 gpurun_out/q.err
python - <<P
import json
d=json.load(open('gpurun_out/q.json'))
r=d['roofline']; o=d['other_workloads']
f=lambda m:{k:round(v,3) for k,v in m.items()}
print('value %.2fM'%(d['value']/1e6), 'pipelined', f(r['stage_ms']), 'serial', f(r['serial_stage_ms']), 'frac %.3f serial_frac %.3f'%(r['frac'], r['serial_frac']))
for k,v in o.items(): print(k, '%.2fM'%(v['frames_per_s']/1e6), f(v['stage_ms']), f(v.get('serial_stage_ms',{})))
P
