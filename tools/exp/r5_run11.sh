#!/bin/bash
# round 5, run 11: the frame-parallel SBC rewrite (k_sbc_frames / plan / par / gen): parity suite, then timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5l
timeout 400 python -m pytest tests/test_gpu_sbc.py -x -q > gpurun_out/r5l/sbc_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5l/sbc_tests.log
timeout 200 python tools/exp/r5_sbc.py > gpurun_out/r5l/sbc_time.json 2> gpurun_out/r5l/sbc_time.err; echo "time rc=$?"; cat gpurun_out/r5l/sbc_time.json; tail -5 gpurun_out/r5l/sbc_time.err
