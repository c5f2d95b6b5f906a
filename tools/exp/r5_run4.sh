#!/bin/bash
# round 5, GPU call 4: k_recon_all with the three-deep request pipeline and stream-major item order; new tests; bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
( time EFX_CHECK_LIBS=s,w,t,e timeout 600 python tools/exp/r5_recon_check.py > $O/recon_check.jsonl 2> $O/recon_check.err ) 2> $O/recon_check.time; echo "check rc=$?" >> $O/rc.txt
MODE=2
grep -q '"ALL_OK": true' $O/recon_check.jsonl || MODE=0
echo "tests run with EFX_RECON_MODE=$MODE" >> $O/rc.txt
EFX_RECON_MODE=$MODE timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -5 $O/pytest.log
EFX_RECON_MODE=$MODE timeout 600 python bench.py > $O/bench_mode$MODE.json 2> $O/bench_mode$MODE.err; echo "bench mode$MODE rc=$?" >> $O/rc.txt
python tools/bench_video.py > $O/bench_video.jsonl 2> $O/bench_video.err; echo "bench_video rc=$?" >> $O/rc.txt
cat $O/rc.txt
