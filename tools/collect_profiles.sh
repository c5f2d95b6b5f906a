#!/bin/bash
# Collects the rocprofv3 material that tools/summarize_profiles.py condenses into profiles/:
# kernel-trace stats of the default (pipelined) bench command and of the serial one, then separate
# --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ).  Every pass is bounded by `timeout`.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$1; mkdir -p $out
# --timed-only + 200 steps: 2400 of the 2436 k_recon launches of the process (12 per step: one group of 1024 streams x 12
# picture indexes) belong to the timed region, so the average rocprofv3 reports is the one bench.py measures with HIP
# events (roofline.avg_launch_ms)
B="python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --timed-only"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipelined -- $B > $out/bench_pipelined.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o serial -- $B --no-overlap > $out/bench_serial.log 2>&1
S="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --no-overlap"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- $S > /dev/null 2>&1
done
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_SQ -o p -- $S > /dev/null 2>&1
# the video-out kernels (BASELINE configs[3]): kernel-trace stats and FETCH / WRITE passes of tools/bench_video.py
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o video -- python tools/bench_video.py > $out/bench_video.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $out/vpmc_$c -o p -- python tools/bench_video.py > /dev/null 2>&1
done
grep "^{" $out/bench_pipelined.log | cut -c1-200; ls $out
