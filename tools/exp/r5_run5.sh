#!/bin/bash
# round 5, GPU call 5: residency sweep of the persistent reconstruction grid; tests; bench with the best setting
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout 400 python tools/exp/r5_sweep.py 0 > $O/sweep_gop12.jsonl 2> $O/sweep_gop12.err; echo "sweep rc=$?" >> $O/rc.txt
timeout 400 python tools/exp/r5_sweep.py 36 > $O/sweep_wide.jsonl 2> $O/sweep_wide.err; echo "sweep wide rc=$?" >> $O/rc.txt
ITEMS=$(tail -1 $O/sweep_gop12.jsonl | python -c "import json,sys; print(json.loads(sys.stdin.read())['best']['items_per_wave'])")
WAVES=$(tail -1 $O/sweep_gop12.jsonl | python -c "import json,sys; print(json.loads(sys.stdin.read())['best']['waves_per_cu'])")
echo "best: items $ITEMS waves $WAVES" >> $O/rc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -3 $O/pytest.log
EFX_RECON_ITEMS=$ITEMS EFX_RECON_WAVES=$WAVES timeout 600 python bench.py > $O/bench_best.json 2> $O/bench_best.err; echo "bench best rc=$?" >> $O/rc.txt
EFX_RECON_MODE=0 timeout 600 python bench.py --no-cpu-baseline --no-video-out > $O/bench_mode0.json 2> $O/bench_mode0.err; echo "bench mode0 rc=$?" >> $O/rc.txt
cat $O/rc.txt
