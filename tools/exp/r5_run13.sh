#!/bin/bash
# round 5, run 13: per-kernel times of the SBC pipeline, mono clean and mono mixed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5n
for cfg in mono_clean mono_mixed; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5n/prof_$cfg -o sbc -- python tools/exp/r5_sbc.py $cfg > gpurun_out/r5n/$cfg.json 2>/dev/null
  f=$(find gpurun_out/r5n/prof_$cfg -name "*kernel_stats.csv" | head -1)
  python - "$f" > gpurun_out/r5n/${cfg}_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"].split("(")[0]:34s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.2f} min_us {float(r["MinNs"])/1e3:9.2f}')
PY
  cat gpurun_out/r5n/$cfg.json; cat gpurun_out/r5n/${cfg}_kernels.txt
  rm -rf gpurun_out/r5n/prof_$cfg
done
