#!/bin/bash
# development aid: extra PMC passes for k_recon / k_parse (each pass bounded by `timeout`; rocprofv3
# hangs in its signal handler when a counter set exceeds the hardware slots)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 90 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcx_$name -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/pmcx_$name.log 2>&1; echo "$name rc=$?"; }
run ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
run sq SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
run spi SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WAVE
run grbm GRBM_GUI_ACTIVE GRBM_SPI_BUSY
