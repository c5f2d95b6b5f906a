#!/bin/bash
# round 5, run 27+: SBC tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5t31
timeout 600 python -m pytest tests/test_gpu_sbc.py -x -q > gpurun_out/r5t31/sbc.log 2>&1; echo "sbc tests rc=$?"; tail -12 gpurun_out/r5t31/sbc.log | cut -c1-600
timeout 200 python tools/exp/r5_sbc.py > gpurun_out/r5t31/sbc_time.json 2>/dev/null; cat gpurun_out/r5t31/sbc_time.json
