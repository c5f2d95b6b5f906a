#!/bin/bash
# round 5, GPU call 8: the final tree -- build check, smoke, -m gpu, default bench (twice)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.txt
tail -2 $O/pytest.log
for i in 1 2; do
  timeout 600 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?" >> $O/rc.txt
done
timeout 300 python bench.py --pin-parse-cap --no-fixed-batch --no-other-workloads --no-video-out --no-cpu-baseline --sustained-steps 0 > $O/bench_pincap.json 2> $O/bench_pincap.err; echo "bench pincap rc=$?" >> $O/rc.txt
cat $O/rc.txt
