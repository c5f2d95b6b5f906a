#!/bin/bash
# round 5, run 25: the slot-table allocation fixed: SBC tests (with the two guard-allocator cases), then demux / edge under both guards
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5guard2
timeout 600 python -m pytest tests/test_gpu_sbc.py -x -q > gpurun_out/r5guard2/sbc.log 2>&1; echo "sbc tests rc=$?"; tail -6 gpurun_out/r5guard2/sbc.log | cut -c1-400
for g in 1 2; do
  EFX_GUARD=$g timeout 600 python -m pytest tests/test_gpu_sbc.py tests/test_gpu_demux.py tests/test_gpu_edge.py -x -q > gpurun_out/r5guard2/tests_$g.log 2>&1; echo "guard $g tests rc=$?"; tail -2 gpurun_out/r5guard2/tests_$g.log | cut -c1-300
done
