for spec in "" "EFX_RECON2=1" "EFX_RECON2=1 EFX_PARSE_WG_CAP=224" "EFX_RECON2=1 EFX_PARSE_WG_CAP=288" "EFX_PARSE_WG_CAP=224" "EFX_RECON2=1 EFX_PARSE_WG_CAP=0"; do EFX_LIB=$GRAFT_REPO_ROOT/espflix_amd/libefx_ps1.so python tools/exp/env_sweep.py "$spec"; done
echo base; python tools/exp/env_sweep.py ""
echo hwq; GPU_MAX_HW_QUEUES=8 python tools/exp/env_sweep.py "EFX_RECON2=1" ""
