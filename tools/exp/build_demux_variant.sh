#!/bin/bash
# usage: build_demux_variant.sh tag "flags"  -> espflix_amd/libefx_<tag>.so (k_demux.hip and efx_api.hip rebuilt with the flags)
set -e
tag=$1; flags=$2
d=/tmp/var_$tag; mkdir -p $d
cd /root/repo
for f in efx_api k_demux; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iespflix_amd/csrc $flags -c espflix_amd/csrc/$f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/efx_api.o $d/k_demux.o espflix_amd/csrc/k_sbc.o espflix_amd/csrc/k_parse.o espflix_amd/csrc/k_recon.o espflix_amd/csrc/k_index.o espflix_amd/csrc/k_video.o espflix_amd/csrc/k_tsindex.o espflix_amd/csrc/efx_tables.o espflix_amd/csrc/efx_multi.o -o espflix_amd/libefx_$tag.so
echo built $tag
