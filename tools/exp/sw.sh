L=$GRAFT_REPO_ROOT/espflix_amd
for t in a b a b; do EFX_LIB=$L/libefx_$t.so python tools/exp/env_sweep.py "T=$t"; done
for t in a b; do FLAGS=36 EFX_LIB=$L/libefx_$t.so python tools/exp/env_sweep.py "wide=$t"; done
EFX_LIB=$L/libefx_pb.so timeout 200 python tools/dbg/probe_waves.py pipelined 2>&1 | grep -A6 "^k_parse"
EFX_LIB=$L/libefx_b.so timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_edge.py -m gpu -x -q 2>&1 | tail -2
