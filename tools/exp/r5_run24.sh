#!/bin/bash
# round 5, run 24: the new SBC and demux kernels under the guard-page allocator (both alignments): their tests, then the
# video_out leg of bench.py five times per alignment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5guard
for g in 1 2; do
  EFX_GUARD=$g timeout 600 python -m pytest tests/test_gpu_sbc.py tests/test_gpu_demux.py tests/test_gpu_edge.py -x -q > gpurun_out/r5guard/tests_$g.log 2>&1; echo "guard $g tests rc=$?"; tail -2 gpurun_out/r5guard/tests_$g.log
  EFX_GUARD=$g timeout 600 python bench.py --soak video_out 5 > gpurun_out/r5guard/soak_$g.log 2>&1; echo "guard $g soak rc=$?"; tail -2 gpurun_out/r5guard/soak_$g.log | cut -c1-300
done
