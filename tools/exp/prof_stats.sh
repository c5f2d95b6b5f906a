cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r2b; mkdir -p $out
B="python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-fixed-batch --no-other-workloads --timed-only"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipelined -- $B > $out/bench_pipelined.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o serial -- $B --no-overlap > $out/bench_serial.log 2>&1
grep "^{" $out/bench_pipelined.log | cut -c1-120
