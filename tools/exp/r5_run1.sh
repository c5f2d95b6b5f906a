#!/bin/bash
# round 5, GPU call 1: reconstruction launch structures (parity + timing), the -m gpu suite, guard-page soak of the r4 code path
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
rocminfo 2>/dev/null | grep -m1 -i "gfx950" > $O/box.txt
( time timeout 400 python tools/exp/r5_recon_check.py > $O/recon_check_a.jsonl 2> $O/recon_check_a.err ) 2> $O/recon_check_a.time; echo "check_a rc=$?" >> $O/rc.txt
( time EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_b.so timeout 300 python tools/exp/r5_recon_check.py quick > $O/recon_check_b.jsonl 2> $O/recon_check_b.err ) 2> $O/recon_check_b.time; echo "check_b rc=$?" >> $O/rc.txt
MODE=2
grep -q '"ALL_OK": true' $O/recon_check_a.jsonl || MODE=0
echo "tests run with EFX_RECON_MODE=$MODE" >> $O/rc.txt
EFX_RECON_MODE=$MODE timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -5 $O/pytest.log
# guard-page soak: first the launch structure that faulted once in round 4 (one k_recon launch per picture index)
for leg in primary:25 wide1500k:12 vmedia_x1024:12 video_out:4 fixed_batch_8192:4; do
  l=${leg%%:*}; n=${leg##*:}
  EFX_GUARD=1 EFX_RECON_MODE=0 timeout 400 python bench.py --soak $l $n > $O/soak_m0_$l.out 2> $O/soak_m0_$l.err; echo "soak mode0 guard1 $l x$n rc=$?" >> $O/rc.txt
  tail -2 $O/soak_m0_$l.err
done
cat $O/rc.txt
