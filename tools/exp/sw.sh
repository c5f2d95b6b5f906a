L=$GRAFT_REPO_ROOT/espflix_amd
for t in a pad256 pad640 pad1280 pad2304 pad4352 a; do EFX_LIB=$L/libefx_$t.so python tools/exp/env_sweep.py "T=$t"; done
