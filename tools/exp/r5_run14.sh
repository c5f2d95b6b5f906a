#!/bin/bash
# round 5, run 14: SBC variants (sample stage / matrixing stage / residency), mono clean, kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5r
for v in "" k1 k2 k4 k7; do
  L=$GRAFT_REPO_ROOT/espflix_amd/libefx.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  EFX_LIB=$L timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5r/prof_$v -o sbc -- python tools/exp/r5_sbc.py mono_clean > gpurun_out/r5r/t_$v.json 2>/dev/null
  f=$(find gpurun_out/r5r/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-default} $(cut -c1-110 gpurun_out/r5r/t_$v.json)"; python tools/exp/kstats.py $f | grep "par_mono\|frames"
  rm -rf gpurun_out/r5r/prof_$v
done 2>&1 | tee gpurun_out/r5r/variants.txt
