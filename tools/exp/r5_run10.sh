#!/bin/bash
# round 5, GPU call 10: k_demux gather with batched searches, staged in LDS (default) against straight from global memory
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
for lib in "" d0; do
  L=""; [ -n "$lib" ] && L="EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$lib.so"
  env $L timeout 600 python -m pytest tests/test_gpu_demux.py tests/test_gpu_edge.py tests/test_gpu_sbc.py tests/test_gpu_decode.py -x -q > $O/pytest_$lib.log 2>&1; echo "pytest '$lib' rc=$?" >> $O/rc.txt
  tail -1 $O/pytest_$lib.log
  for i in 1 2; do env $L python tools/bench_video.py 2>/dev/null | grep k_demux | cut -c1-200 >> $O/demux_$lib.jsonl; done
  cat $O/demux_$lib.jsonl
done
cat $O/rc.txt
