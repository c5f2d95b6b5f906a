python tools/exp/time_variants.py base prio0 prio1 2>&1 | grep -v amdgpu.ids
for n in 32 64 128; do echo "parse on $n CUs"; EFX_EXP_PARSE_CUS=$n python tools/exp/time_variants.py base 2>&1 | grep -v amdgpu.ids; echo "parse on $n CUs, recon on the rest"; EFX_EXP_PARSE_CUS=$n EFX_EXP_RECON_REST=1 python tools/exp/time_variants.py base 2>&1 | grep -v amdgpu.ids; done
