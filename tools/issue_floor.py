#!/usr/bin/env python3
"""issue_floor.py -- the vector-instruction ISSUE FLOOR of the decoder's kernels, with ONE convention (round 6).

Convention (profiles/r6_valu_rates.md, measured by tools/exp/valu_rates.hip on gfx950):

    a SIMD of a CU issues one wave64 vector instruction per  `cycles(opcode)`  core cycles, 2.2 for the FULL-rate opcodes
    (v_add / v_sub / v_and / v_or / v_xor / v_not / v_mov / v_lshrrev / v_ashrrev / v_bitop3 / v_fma_f32 ...) and 4.2 for
    everything else this path uses (v_mad_i32_i24, v_mul_i32_i24, v_perm_b32, v_lerp_u8, v_pk_*16, v_bfi, v_cndmask, v_cmp,
    v_lshlrev, v_min / v_max, v_bfe, DPP / SDWA forms, any VOP2 with an SGPR operand is measured too but not tracked here).

    issue floor of a launch = dynamic vector instructions (SQ_INSTS_VALU, rocprofv3) x the kernel's mix-weighted cycles per
    instruction / (1024 SIMDs x clock).  The mix is the STATIC opcode histogram of the compiled kernel (hipcc -S): these
    kernels are straight-line code with short uniform loops, and the static count of k_recon (1659) is within 9 % of the
    dynamic count per wave (1526).

    python tools/issue_floor.py [--rates profiles/r6_valu_rates.txt] [--pmc profiles/r6_pmc_summary.json] [--out profiles/r6_issue_floor.json]

Needs hipcc (cross-compiles without a GPU).  bench.py reads the JSON this writes.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "espflix_amd", "csrc")
CLOCK_HZ = 2.4e9   # MI355X_MICROARCH.md: max clock; the measured clock under this load is 2.2-2.4 GHz (r6_valu_rates.txt, MHz column)
SIMDS = 256 * 4

# kernel (mangled-name fragment) -> (source file, extra flags)
KERNELS = {
    "k_recon": ("k_recon.hip", "_ZN3efx7k_reconE", ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]),
    "k_parse": ("k_parse.hip", "_ZN3efx7k_parseE", []),
    "k_sbc_par_mono": ("k_sbc.hip", "_ZN3efx14k_sbc_par_monoE", []),
    "k_pdm": ("k_video.hip", "_ZN3efx5k_pdmE", []),
    "k_composite": ("k_video.hip", "_ZN3efx11k_compositeE", []),
}


def read_rates(path):
    """opcode -> issue cycles per wave64 instruction per SIMD at 8 waves per SIMD (the `issue cyc` column)."""
    rates = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 6 and f[1] == "8" and f[2] == "8":
            try:
                rates[f[0]] = float(f[4])
            except ValueError:
                pass
    return rates


def classify(op, rates):
    """(cycles, how) for a compiled opcode."""
    if op.endswith("_dpp"):
        return rates.get("v_add_u32_dpp_row_shr", 4.2), "dpp"
    if op.endswith("_sdwa"):
        return rates.get("v_add_u32_sdwa", 4.2), "sdwa"
    base = re.sub(r"_(e32|e64)$", "", op)
    alias = {"v_cmp": "v_cmp_lt_i32", "v_cndmask_b32": "v_cndmask_b32_sgpr", "v_mul_hi_i32": "v_mul_hi_u32", "v_min_u32": "v_min_i32",
             "v_max_u32": "v_max_i32", "v_min_i32": "v_min_i32", "v_fmac_f32": "v_mac_f32", "v_bitop3_b16": "v_bitop3_b32",
             "v_lshl_add_u64": "v_lshl_add_u64", "v_readlane_b32": "v_readlane_b32", "v_readfirstlane_b32": "v_readfirstlane_b32",
             "v_cvt_f32_u32": "v_cvt_f32_i32", "v_cvt_u32_f32": "v_cvt_f32_i32", "v_rcp_iflag_f32": "v_rcp_f32", "v_mbcnt_lo_u32_b32": "v_add3_u32",
             "v_mbcnt_hi_u32_b32": "v_add3_u32", "v_ffbh_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_i32", "v_sub_co_u32": "v_add_co_u32",
             "v_addc_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_add_co_u32", "v_subbrev_co_u32": "v_add_co_u32", "v_mul_u32_u24": "v_mul_u32_u24",
             "v_writelane_b32": "v_readlane_b32", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32", "v_mov_b64": "v_pk_fma_f32",
             "v_lshlrev_b64": "v_lshl_add_u64", "v_lshrrev_b64": "v_lshl_add_u64", "v_ashrrev_i64": "v_lshl_add_u64", "v_pk_mov_b32": "v_pk_fma_f32",
             "v_mad_u64_u32": "v_mad_u64_u32", "v_min3_i32": "v_max3_i32", "v_min3_u32": "v_max3_i32", "v_max3_u32": "v_max3_i32",
             "v_med3_u32": "v_med3_i32", "v_sub_u16": "v_mad_i16", "v_add_u16": "v_mad_i16", "v_subrev_u32": "v_subrev_u32"}
    if base.startswith("v_cmp_") or base.startswith("v_cmpx_"):
        base = "v_cmp"
    name = alias.get(base, base)
    if name in rates:
        return rates[name], "measured" if name == base else "as " + name
    return None, "unmeasured"


def histogram(src, frag, flags):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}", *flags,
               "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        hist = collections.Counter()
        inside = False
        for line in open(out):
            if line.startswith(frag) and ":" in line.split(";")[0]:
                inside = True
                continue
            if not inside:
                continue
            t = line.strip()
            if t.startswith("s_endpgm"):
                break
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            hist[t.split()[0]] += 1
        return hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r6_valu_rates.txt"))
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_issue_floor.json"))
    a = ap.parse_args()
    rates = read_rates(a.rates)
    pmc = {}
    for name in ([a.pmc] if a.pmc else []) + [os.path.join(ROOT, "profiles", n) for n in ("r6_pmc_summary.json", "r5_pmc_summary.json")]:
        if name and os.path.exists(name):
            doc = json.load(open(name))
            for key in ("kernels", "video_kernels"):
                for k, v in doc.get(key, {}).items():
                    pmc.setdefault(k.replace("efx::", ""), v)
    doc = {"convention": "issue cycles per wave64 vector instruction per SIMD, measured at 8 waves per SIMD (tools/exp/valu_rates.hip -> "
                         + os.path.relpath(a.rates, ROOT) + "); floor = SQ_INSTS_VALU x mix-weighted cycles / (1024 SIMDs x 2.4 GHz)",
           "full_rate_cycles": rates.get("v_add_u32"), "half_rate_cycles": rates.get("v_mad_i32_i24"), "clock_hz": CLOCK_HZ, "kernels": {}}
    for k, (src, frag, flags) in KERNELS.items():
        hist = histogram(src, frag, flags)
        valu = {op: n for op, n in hist.items() if op.startswith("v_")}
        n_valu = sum(valu.values())
        cyc = 0.0
        unmeasured = {}
        by_class = collections.Counter()
        for op, n in valu.items():
            c, how = classify(op, rates)
            if c is None:
                unmeasured[op] = n
                c = rates.get("v_mad_i32_i24", 4.2)
            cyc += c * n
            by_class["full" if c < 3 else ("half" if c < 6 else "slower")] += n
        ent = {"static_valu": n_valu, "static_lds": sum(n for op, n in hist.items() if op.startswith("ds_")),
               "static_vmem": sum(n for op, n in hist.items() if op.startswith(("global_", "buffer_", "flat_", "scratch_"))),
               "static_salu": sum(n for op, n in hist.items() if op.startswith("s_")),
               "by_rate": dict(by_class), "cycles_per_valu": cyc / max(1, n_valu), "static_issue_cycles": cyc,
               "unmeasured_priced_half_rate": unmeasured, "top": dict(collections.Counter(valu).most_common(14))}
        p = pmc.get(k)
        if p and p.get("SQ_INSTS_VALU"):
            dyn = p["SQ_INSTS_VALU"]
            ent["dynamic_valu_per_launch"] = dyn
            ent["waves_per_launch"] = p.get("SQ_WAVES")
            ent["issue_floor_us"] = dyn * ent["cycles_per_valu"] / (SIMDS * CLOCK_HZ) * 1e6
        doc["kernels"][k] = ent
        print("%-16s static VALU %5d (full %d / half %d / slower %d)  %.2f cycles per instruction%s" % (
            k, n_valu, by_class["full"], by_class["half"], by_class["slower"], ent["cycles_per_valu"],
            "  floor %.1f us per launch (%.1f M dynamic)" % (ent["issue_floor_us"], ent["dynamic_valu_per_launch"] / 1e6) if "issue_floor_us" in ent else ""))
        if unmeasured:
            print("                 unmeasured, priced at the half rate:", unmeasured)
    json.dump(doc, open(a.out, "w"), indent=1, sort_keys=True)
    print("wrote", os.path.relpath(a.out, ROOT))


if __name__ == "__main__":
    main()
