#!/bin/bash
# round 5, run 21: k_demux chunk size / workgroup size variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5do
for v in "" d16t128 d16t64 d32t64 d12t64 d20t64 d24t64; do
  L=$GRAFT_REPO_ROOT/espflix_amd/libefx.so; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  echo "== ${v:-default} $(EFX_LIB=$L timeout 120 python tools/exp/r5_demux.py 2>/dev/null)"
done | tee gpurun_out/r5do/variants.txt
