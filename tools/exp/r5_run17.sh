#!/bin/bash
# round 5, run 17: SBC with MSB-first staging and two-row matrixing (default lib: chunk 16); tests + timings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5ae
timeout 300 python -m pytest tests/test_gpu_sbc.py -x -q > gpurun_out/r5ae/sbc_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5ae/sbc_tests.log
for v in ""; do
  L=$GRAFT_REPO_ROOT/espflix_amd/libefx.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  EFX_LIB=$L timeout 100 python tools/exp/r5_sbc.py > gpurun_out/r5ae/all_$v.json 2>/dev/null; cat gpurun_out/r5ae/all_$v.json
  EFX_LIB=$L timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5ae/prof_$v -o sbc -- python tools/exp/r5_sbc.py mono_clean > gpurun_out/r5ae/t_$v.json 2>/dev/null
  f=$(find gpurun_out/r5ae/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-default} $(cut -c1-110 gpurun_out/r5ae/t_$v.json)"; python tools/exp/kstats.py $f | grep "k_sbc_par_mono\|k_sbc_frames"
  rm -rf gpurun_out/r5ae/prof_$v
done 2>&1 | tee gpurun_out/r5ae/variants.txt
