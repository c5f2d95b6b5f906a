#!/bin/bash
# development aid: kernel-trace stats of the libefx builds named on the command line (libefx_<x>.so), same box
for v in "$@"; do
  export EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_$v.so
  echo "== $v"; bash tools/exp/stats_quick.sh ab_$v --no-overlap | grep -E "k_recon|k_parse|k_classify"
done
