#!/bin/bash
# round 5, run 19: per-kernel times of a mixed SBC batch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5af
for cfg in mono_mixed; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5af/prof_$cfg -o sbc -- python tools/exp/r5_sbc.py $cfg > gpurun_out/r5af/$cfg.json 2>/dev/null
  f=$(find gpurun_out/r5af/prof_$cfg -name "*kernel_stats.csv" | head -1)
  cat gpurun_out/r5af/$cfg.json; python tools/exp/kstats.py $f | grep k_sbc_ | tee gpurun_out/r5af/${cfg}_kernels.txt
  rm -rf gpurun_out/r5af/prof_$cfg
done
