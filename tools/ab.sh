#!/bin/bash
# development aid: A/B two builds of libefx on the same box (EFX_LIB selects the library)
for rep in 1 2 3; do
  for v in a b; do
    EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['serial_stage_ms'].items()})"
  done
done
