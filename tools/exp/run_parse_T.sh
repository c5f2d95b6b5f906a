python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for T in 16,32 1,1 8,16 24,48 32,32 48,48; do
  EFX_PARSE_T=$T python bench.py --steps 20 --warmup 5 --no-fixed-batch --no-cpu-baseline > gpurun_out/t_$T.json 2> gpurun_out/t_$T.err || tail -3 gpurun_out/t_$T.err
  python - <<P
import json
d=json.load(open('gpurun_out/t_$T.json'))
r=d['roofline']; o=d['other_workloads']
print('$T', 'value %.2fM'%(d['value']/1e6), 'serial', {k:round(v,3) for k,v in r['serial_stage_ms'].items()}, 'wide %.2fM'%(o['wide_slices_1500k']['frames_per_s']/1e6), {k:round(v,3) for k,v in o['wide_slices_1500k']['serial_stage_ms'].items()}, 'vmedia %.2fM'%(o['vmedia_x1024']['frames_per_s']/1e6))
P
done
