#!/bin/bash
# development aid (round 6): the vector-memory counters of k_recon for libefx_<tag>.so builds, one call at a time, per launch of
# 1024 streams (means over the launches of the run): usage  pmc_variants.sh tag [tag ...]   ("lib" = espflix_amd/libefx.so)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="python tools/exp/decode_loop.py 3"
for tag in "$@"; do
  lib=$GRAFT_REPO_ROOT/espflix_amd/libefx_$tag.so; [ "$tag" = lib ] && lib=$GRAFT_REPO_ROOT/espflix_amd/libefx.so
  k=0
  for grp in "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    k=$((k+1))
    EFX_LIB=$lib EFX_FORCE_PARSE_CAP=1 timeout 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pv_${tag}_$k -o p -- $S > gpurun_out/pv_${tag}_$k.log 2>&1 < /dev/null
  done
  python - <<P
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pv_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('efx::k_recon('):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$tag', {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(agg.items())}, 'launches', len(next(iter(agg.values()), [])))
P
  rm -rf gpurun_out/pv_${tag}_?
done
