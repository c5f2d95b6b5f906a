#!/bin/bash
# round 5, GPU call 3: k_recon_all with its hand-over words on separate lines; chunk-parallel k_demux; full bench line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python tools/guard_selftest.py > $O/guard_selftest.txt 2>&1; echo "guard_selftest rc=$?" >> $O/rc.txt
( time EFX_CHECK_LIBS=s,b,g,e timeout 600 python tools/exp/r5_recon_check.py > $O/recon_check.jsonl 2> $O/recon_check.err ) 2> $O/recon_check.time; echo "check rc=$?" >> $O/rc.txt
MODE=2
grep -q '"ALL_OK": true' $O/recon_check.jsonl || MODE=0
echo "tests run with EFX_RECON_MODE=$MODE" >> $O/rc.txt
EFX_RECON_MODE=$MODE timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -5 $O/pytest.log
for m in 0 2; do
  EFX_RECON_MODE=$m timeout 600 python bench.py > $O/bench_mode$m.json 2> $O/bench_mode$m.err; echo "bench mode$m rc=$?" >> $O/rc.txt
done
for cfg in "1:$MODE:primary:60" "1:$MODE:wide1500k:40" "1:$MODE:vmedia_x1024:20" "1:$MODE:video_out:10" "1:$MODE:fixed_batch_8192:6"; do
  IFS=: read g m l n <<< "$cfg"
  EFX_GUARD=$g EFX_RECON_MODE=$m timeout 600 python bench.py --soak $l $n > $O/soak_g${g}_m${m}_$l.out 2> $O/soak_g${g}_m${m}_$l.err; echo "soak guard$g mode$m $l x$n rc=$?" >> $O/rc.txt
  tail -1 $O/soak_g${g}_m${m}_$l.err
done
cat $O/rc.txt
