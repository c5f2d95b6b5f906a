#!/bin/bash
# round 5, GPU call 6: final structure (one k_recon launch per picture index, three bitstream buffers): tests, bench, soak top-up
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
for cfg in "1:wide1500k:30" "1:vmedia_x1024:30" "1:video_out:86" "1:fixed_batch_8192:88" "2:primary:60" "2:wide1500k:40"; do
  IFS=: read g l n <<< "$cfg"
  EFX_GUARD=$g timeout 600 python bench.py --soak $l $n > $O/soak_g${g}_$l.out 2> $O/soak_g${g}_$l.err; echo "soak guard$g $l x$n rc=$?" >> $O/rc.txt
  tail -1 $O/soak_g${g}_$l.err
done
cat $O/rc.txt
