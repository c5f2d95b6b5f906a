#!/bin/bash
# round 5, GPU call 2: guard self-test; reconstruction variants (sc1 load / store forms) side by side; tests; guard soak
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
timeout 300 python tools/guard_selftest.py > $O/guard_selftest.txt 2>&1; echo "guard_selftest rc=$?" >> $O/rc.txt
( time EFX_CHECK_LIBS=b,c,d,e,g,f timeout 600 python tools/exp/r5_recon_check.py > $O/recon_check.jsonl 2> $O/recon_check.err ) 2> $O/recon_check.time; echo "check rc=$?" >> $O/rc.txt
MODE=2
grep -q '"ALL_OK": true' $O/recon_check.jsonl || MODE=0
echo "tests run with EFX_RECON_MODE=$MODE" >> $O/rc.txt
EFX_RECON_MODE=$MODE timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -5 $O/pytest.log
for cfg in "1:0:primary:100" "1:0:wide1500k:60" "1:0:vmedia_x1024:60" "1:0:video_out:12" "1:0:fixed_batch_8192:10" \
           "2:0:primary:40" "2:0:wide1500k:20" "2:0:vmedia_x1024:20" "2:0:video_out:6" "2:0:fixed_batch_8192:4" \
           "1:$MODE:primary:40" "1:$MODE:wide1500k:20" "1:$MODE:vmedia_x1024:20" "1:$MODE:fixed_batch_8192:4"; do
  IFS=: read g m l n <<< "$cfg"
  EFX_GUARD=$g EFX_RECON_MODE=$m timeout 600 python bench.py --soak $l $n > $O/soak_g${g}_m${m}_$l.out 2> $O/soak_g${g}_m${m}_$l.err; echo "soak guard$g mode$m $l x$n rc=$?" >> $O/rc.txt
  tail -1 $O/soak_g${g}_m${m}_$l.err
done
cat $O/rc.txt
