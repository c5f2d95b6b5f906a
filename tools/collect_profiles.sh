#!/bin/bash
# Collects the rocprofv3 material that tools/summarize_profiles.py condenses into profiles/:
# kernel-trace stats of the default (pipelined) bench command and of the serial one, then separate
# --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ; TA / TCP / LDS for k_recon).  Every pass is bounded by `timeout`.
# Round 5: the PMC passes run one call at a time (rocprofv3 serialises dispatches while it counts), which used to mean an
# UNCAPPED k_parse -- not the kernel the timed region runs.  EFX_FORCE_PARSE_CAP=1 pins the cap (efx_set_option's
# environment default), so both schedules are collected: pmc_* = as shipped (capped), pmcu_* = uncapped.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$1; mkdir -p $out
# --timed-only + 200 steps: 2400 of the 2436 k_recon launches of the process (12 per step: one group of 1024 streams x 12
# picture indexes, pinned) belong to the timed region, so the average rocprofv3 reports is the one bench.py measures with HIP
# events (roofline.avg_launch_ms)
B="python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --timed-only"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipelined -- $B > $out/bench_pipelined.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o serial -- $B --no-overlap > $out/bench_serial.log 2>&1
S="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-batch --no-other-workloads --no-video-out --no-overlap"
for c in FETCH_SIZE WRITE_SIZE; do
  EFX_FORCE_PARSE_CAP=1 timeout 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- $S > /dev/null 2>&1
  EFX_FORCE_PARSE_CAP=2 timeout 120 rocprofv3 --pmc $c --output-format csv -d $out/pmcu_$c -o p -- $S > /dev/null 2>&1
done
EFX_FORCE_PARSE_CAP=1 timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_SQ -o p -- $S > /dev/null 2>&1
# what a CU's resident k_recon waves queue for: texture-addresser / L1 / LDS busy and stall counters, at the shipped LDS
# footprint (18 waves per CU) and -- EFX_LIB = a build with padded LDS -- at 14
for lib in "" 14; do
  L=""; [ -n "$lib" ] && L="EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_p$lib.so"
  k=0
  for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    k=$((k+1))
    env $L EFX_FORCE_PARSE_CAP=1 timeout 120 rocprofv3 --pmc $grp --output-format csv -d $out/pmcq${lib}_$k -o p -- $S > $out/pmcq${lib}_$k.log 2>&1
  done
done
# the video-out kernels (BASELINE configs[3]): kernel-trace stats and FETCH / WRITE passes of tools/bench_video.py
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o video -- python tools/bench_video.py > $out/bench_video.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $out/vpmc_$c -o p -- python tools/bench_video.py > /dev/null 2>&1 < /dev/null
done
# (round 6) the instruction counts of the video / audio kernels, for their issue floors (tools/issue_floor.py)
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $out/vpmc_SQ -o p -- python tools/bench_video.py > /dev/null 2>&1 < /dev/null
# The merge back from the GPU box is limited to 64 MiB and the per-dispatch counter tables are 200 MB: they are condensed HERE
# (tools/summarize_profiles.py, the same script that writes profiles/) and only the summary, the stats tables and the logs travel.
python tools/summarize_profiles.py $1 --out $out/summary --stats $out/pipelined $out/serial $out/video \
    --pmc $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ --video-pmc $out/vpmc_FETCH_SIZE $out/vpmc_WRITE_SIZE $out/vpmc_SQ \
    --extra kernels_uncapped_parse=$out/pmcu_FETCH_SIZE,$out/pmcu_WRITE_SIZE \
            k_recon_queueing_18_waves_per_cu=$out/pmcq_1,$out/pmcq_2,$out/pmcq_3,$out/pmcq_4 \
            k_recon_queueing_14_waves_per_cu=$out/pmcq14_1,$out/pmcq14_2,$out/pmcq14_3,$out/pmcq14_4 \
    --note "round 6 ($(date -u +%Y-%m-%d), box $(hostname)): one MI355X, 1024 streams x GOP 12; kernels = PMC passes with the parse cap pinned (EFX_FORCE_PARSE_CAP=1: the schedule the timed region runs), kernels_uncapped_parse = the same one call at a time without it; k_recon_queueing_* = TA / TCP / SQ / LDS counters of k_recon per launch of 1024 streams at the shipped LDS footprint and padded to 14 waves per CU" > $out/summary.log 2>&1
rm -rf $out/pmc_* $out/pmcu_* $out/pmcq_? $out/pmcq14_? $out/vpmc_* $out/*_kernel_trace.csv $out/*_agent_info.csv
grep "^{" $out/bench_pipelined.log | cut -c1-200; du -sh $out
