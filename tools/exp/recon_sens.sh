#!/bin/bash
# development aid (round 6): sensitivity of a k_recon launch to added work of one kind -- builds libefx_sens<k>_<n>.so
# kind 1: full-rate VALU (v_add_u32), 2: half-rate VALU (v_perm_b32), 3: LDS reads, 4: vector loads that hit the caches
set -e
cd /root/repo
for spec in "$@"; do
  k=${spec%%:*}; n=${spec##*:}
  bash tools/exp/build_variant.sh sens${k}_${n} "-DEFX_RECON_SENS=$k -DEFX_RECON_SENS_N=$n" | tail -1
done
