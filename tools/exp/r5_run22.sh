#!/bin/bash
# round 5, run 22: per-kernel times of the demux (chunk 16 x 64 threads against the default)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5dp
for v in "" d16t64; do
  L=$GRAFT_REPO_ROOT/espflix_amd/libefx.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  EFX_LIB=$L timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5dp/prof_$v -o dm -- python tools/exp/r5_demux.py > gpurun_out/r5dp/t_$v.json 2>/dev/null
  f=$(find gpurun_out/r5dp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-default} $(cat gpurun_out/r5dp/t_$v.json)"; python tools/exp/kstats.py $f | grep "k_demux\|k_index\|copyBuffer"
  rm -rf gpurun_out/r5dp/prof_$v
done 2>&1 | tee gpurun_out/r5dp/kernels.txt
