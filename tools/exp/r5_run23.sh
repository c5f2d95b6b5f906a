#!/bin/bash
# round 5, run 23: demux with 16-packet chunks as the default: the tests that touch it, then the timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5dq
timeout 900 python -m pytest tests/test_gpu_demux.py tests/test_gpu_edge.py tests/test_gpu_sbc.py tests/test_gpu_index.py tests/test_gpu_dropin.py tests/test_gpu_adapter.py -x -q > gpurun_out/r5dq/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5dq/tests.log
timeout 120 python tools/exp/r5_demux.py 2>/dev/null | tee gpurun_out/r5dq/demux.json
