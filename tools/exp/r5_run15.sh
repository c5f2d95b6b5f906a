#!/bin/bash
# round 5, run 15: merged launches + frames loads; chunk variants of the mono kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5t


for v in c16 c16m1 c16k7 c16s0 c16p16; do
  L=$GRAFT_REPO_ROOT/espflix_amd/libefx.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  EFX_LIB=$L timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5t/prof_$v -o sbc -- python tools/exp/r5_sbc.py mono_clean > gpurun_out/r5t/t_$v.json 2>/dev/null
  f=$(find gpurun_out/r5t/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-default} $(cut -c1-110 gpurun_out/r5t/t_$v.json)"; python tools/exp/kstats.py $f | grep "par_mono"
  rm -rf gpurun_out/r5t/prof_$v
done 2>&1 | tee gpurun_out/r5t/variants.txt
