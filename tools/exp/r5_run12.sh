#!/bin/bash
# round 5, run 12+: SBC tests, timings, per-kernel times
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5o
timeout 400 python -m pytest tests/test_gpu_sbc.py -x -q > gpurun_out/r5o/sbc_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5o/sbc_tests.log
timeout 200 python tools/exp/r5_sbc.py > gpurun_out/r5o/sbc_time.json 2> gpurun_out/r5o/sbc_time.err; echo "time rc=$?"; cat gpurun_out/r5o/sbc_time.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5o/prof -o sbc -- python tools/exp/r5_sbc.py mono_clean > /dev/null 2>&1
f=$(ls gpurun_out/r5o/prof/*/sbc_kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find gpurun_out/r5o/prof -name "*kernel_stats.csv" | head -1)
python tools/exp/kstats.py $f | tee gpurun_out/r5o/sbc_kernel_stats.txt
rm -rf gpurun_out/r5o/prof
