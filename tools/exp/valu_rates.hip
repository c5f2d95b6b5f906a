// valu_rates.hip -- what a wave64 instruction of the decoder's arithmetic costs a SIMD of gfx950 to ISSUE.
//
// Development aid (round 6): k_recon / k_parse / k_sbc_par_* are integer VALU code -- v_mad_i32_i24,
// v_perm_b32, v_lerp_u8, packed 16-bit min / max / add, v_bfi_b32, DPP adds, 16-bit LDS reads (the arithmetic of
// /root/reference/src/player.cpp:922-996 and 767-820).  The repository priced them with three different rates; this
// measures them, opcode by opcode, next to v_fma_f32:
//
//   * a workgroup = 256 threads = one wave per SIMD of its CU; W workgroups per CU = W waves per SIMD (grid 256 x W);
//   * a wave runs `iters` trips of an unrolled body of 8 independent chains x 16 = 128 instructions of ONE opcode
//     (asm volatile: nothing is reordered, nothing is inserted), so with W waves a SIMD has 8 W independent
//     instructions to choose from;
//   * per wave s_memtime and s_memrealtime (100 MHz) are read before and after; the host prints, per opcode and W,
//     ns per instruction per SIMD = (longest wave's real time) / (W x instructions per wave), the same in cycles of
//     the s_memtime counter, and the ratio to v_fma_f32 at the same W;
//   * `chains = 1` rows (one dependent chain, one wave per SIMD) give the issue-to-use latency.
//
//   hipcc --offload-arch=gfx950 -O2 tools/exp/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

struct Stamp {
    unsigned long long cyc, real, begin, end;
};

// F_<name>(d): the instruction text for a chain whose register is operand d ("%0" ... "%7"); %8, %9 = two other vector
// registers, %10 = an SGPR, %11 = a 64-bit vector register pair.  A whole loop body -- 16 x 8 instructions -- is ONE asm
// statement: between two asm statements with clobbers the compiler's hazard recognizer puts an `s_nop 0`, which would be
// measured along.
#define F_v_fma_f32(d) "v_fma_f32 " d ", " d ", %8, %9\n"
#define F_v_pk_fma_f32(d) "v_pk_fma_f32 " d ", " d ", %11, %11\n"
#define F_v_mov_b32(d) "v_mov_b32 " d ", %8\n"
#define F_v_add_u32(d) "v_add_u32 " d ", " d ", %8\n"
#define F_v_sub_u32(d) "v_sub_u32 " d ", " d ", %8\n"
#define F_v_add_u32_sgpr(d) "v_add_u32 " d ", %10, " d "\n"
#define F_v_add3_u32(d) "v_add3_u32 " d ", " d ", %8, %9\n"
#define F_v_lshl_add_u32(d) "v_lshl_add_u32 " d ", " d ", 3, %8\n"
#define F_v_and_b32(d) "v_and_b32 " d ", " d ", %8\n"
#define F_v_xor_b32(d) "v_xor_b32 " d ", " d ", %8\n"
#define F_v_and_or_b32(d) "v_and_or_b32 " d ", " d ", %8, %9\n"
#define F_v_or3_b32(d) "v_or3_b32 " d ", " d ", %8, %9\n"
#define F_v_bfi_b32(d) "v_bfi_b32 " d ", %8, " d ", %9\n"
#define F_v_bitop3_b32(d) "v_bitop3_b32 " d ", " d ", %8, %9 bitop3:0x96\n"
#define F_v_bfe_u32(d) "v_bfe_u32 " d ", " d ", 3, 17\n"
#define F_v_bfe_i32(d) "v_bfe_i32 " d ", " d ", 3, 17\n"
#define F_v_lshlrev_b32(d) "v_lshlrev_b32 " d ", 1, " d "\n"
#define F_v_lshrrev_b32(d) "v_lshrrev_b32 " d ", 1, " d "\n"
#define F_v_ashrrev_i32(d) "v_ashrrev_i32 " d ", 1, " d "\n"
#define F_v_ashrrev_i32_v(d) "v_ashrrev_i32 " d ", %8, " d "\n"
#define F_v_alignbit_b32(d) "v_alignbit_b32 " d ", " d ", %8, 8\n"
#define F_v_alignbit_b32_v(d) "v_alignbit_b32 " d ", " d ", %8, %9\n"
#define F_v_perm_b32(d) "v_perm_b32 " d ", " d ", %8, %9\n"
#define F_v_perm_b32_sgpr(d) "v_perm_b32 " d ", " d ", %8, %10\n"
#define F_v_cndmask_b32(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define F_v_cndmask_b32_sgpr(d) "v_cndmask_b32 " d ", " d ", %8, s[22:23]\n"
#define F_v_cndmask_b32_inl(d) "v_cndmask_b32 " d ", 0, " d ", vcc\n"
#define F_v_bitop3_b32_sgpr(d) "v_bitop3_b32 " d ", " d ", %10, %9 bitop3:0x96\n"
#define F_v_and_b32_sgpr(d) "v_and_b32 " d ", %10, " d "\n"
#define F_v_sat_pk_u8_i16(d) "v_sat_pk_u8_i16 " d ", " d "\n"
#define F_v_cndmask_b32_e64_vcc(d) "v_cndmask_b32_e64 " d ", " d ", %8, vcc\n"
#define F_pair_cmp_cndmask_vcc(d) "v_cmp_lt_u32 vcc, " d ", %8\n v_cndmask_b32 " d ", " d ", %9, vcc\n"
#define F_pair_cmp_cndmask_sgpr(d) "v_cmp_lt_u32 s[22:23], " d ", %8\n v_cndmask_b32 " d ", " d ", %9, s[22:23]\n"
#define F_pair_cmp_other_cndmask_vcc(d) "v_cmp_lt_u32 vcc, %8, %9\n v_cndmask_b32 " d ", " d ", %9, vcc\n"
#define F_v_or_b32(d) "v_or_b32 " d ", " d ", %8\n"
#define F_v_not_b32(d) "v_not_b32 " d ", " d "\n"
#define F_v_and_b32_lit(d) "v_and_b32 " d ", 0x7f7f7f7f, " d "\n"
#define F_v_add_u32_inl(d) "v_add_u32 " d ", 7, " d "\n"
#define F_v_subrev_u32(d) "v_subrev_u32 " d ", " d ", %8\n"
#define F_v_add_co_u32(d) "v_add_co_u32 " d ", vcc, " d ", %8\n"
#define F_v_lshlrev_b32_v(d) "v_lshlrev_b32 " d ", %8, " d "\n"
#define F_v_lshrrev_b32_8(d) "v_lshrrev_b32 " d ", 8, " d "\n"
#define F_v_mul_u32_u24(d) "v_mul_u32_u24 " d ", " d ", %8\n"
#define F_v_lshl_or_b32(d) "v_lshl_or_b32 " d ", " d ", 3, %8\n"
#define F_v_add_lshl_u32(d) "v_add_lshl_u32 " d ", " d ", %8, 3\n"
#define F_v_xad_u32(d) "v_xad_u32 " d ", " d ", %8, %9\n"
#define F_v_max3_i32(d) "v_max3_i32 " d ", " d ", %8, %9\n"
#define F_v_lshl_add_u64(d) "v_lshl_add_u64 " d ", " d ", 3, %11\n"
#define F_v_cvt_f32_ubyte0(d) "v_cvt_f32_ubyte0 " d ", " d "\n"
#define F_v_cvt_f32_i32(d) "v_cvt_f32_i32 " d ", " d "\n"
#define F_v_rcp_f32(d) "v_rcp_f32 " d ", " d "\n"
#define F_v_mul_f32(d) "v_mul_f32 " d ", " d ", %8\n"
#define F_v_add_f32(d) "v_add_f32 " d ", " d ", %8\n"
#define F_v_mac_f32(d) "v_fmac_f32 " d ", %8, %9\n"
#define F_v_pk_add_f16(d) "v_pk_add_f16 " d ", " d ", %8\n"
#define F_v_pk_fma_f16(d) "v_pk_fma_f16 " d ", " d ", %8, %9\n"
#define F_v_cmp_lt_i32(d) "v_cmp_lt_i32 vcc, " d ", %8\n"
#define F_v_cmp_lt_i32_sdst(d) "v_cmp_lt_i32 s[20:21], " d ", %8\n"
#define F_v_min_i32(d) "v_min_i32 " d ", " d ", %8\n"
#define F_v_max_i32(d) "v_max_i32 " d ", " d ", %8\n"
#define F_v_med3_i32(d) "v_med3_i32 " d ", " d ", %8, %9\n"
#define F_v_mul_i32_i24(d) "v_mul_i32_i24 " d ", " d ", %8\n"
#define F_v_mul_i32_i24_sgpr(d) "v_mul_i32_i24 " d ", %10, " d "\n"
#define F_v_mad_i32_i24(d) "v_mad_i32_i24 " d ", " d ", %8, %9\n"
#define F_v_mad_i32_i24_sgpr(d) "v_mad_i32_i24 " d ", " d ", %10, %9\n"
#define F_v_mad_u32_u24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n"
#define F_v_mul_lo_u32(d) "v_mul_lo_u32 " d ", " d ", %8\n"
#define F_v_mul_hi_u32(d) "v_mul_hi_u32 " d ", " d ", %8\n"
#define F_v_mul_hi_i32_i24(d) "v_mul_hi_i32_i24 " d ", " d ", %8\n"
#define F_v_mad_u64_u32(d) "v_mad_u64_u32 " d ", s[20:21], %8, %9, " d "\n"
#define F_v_lerp_u8(d) "v_lerp_u8 " d ", " d ", %8, %9\n"
#define F_v_sad_u8(d) "v_sad_u8 " d ", " d ", %8, %9\n"
#define F_v_pk_add_u16(d) "v_pk_add_u16 " d ", " d ", %8\n"
#define F_v_pk_add_i16(d) "v_pk_add_i16 " d ", " d ", %8\n"
#define F_v_pk_sub_i16(d) "v_pk_sub_i16 " d ", " d ", %8\n"
#define F_v_pk_min_i16(d) "v_pk_min_i16 " d ", " d ", %8\n"
#define F_v_pk_max_i16(d) "v_pk_max_i16 " d ", " d ", %8\n"
#define F_v_pk_mul_lo_u16(d) "v_pk_mul_lo_u16 " d ", " d ", %8\n"
#define F_v_pk_mad_i16(d) "v_pk_mad_i16 " d ", " d ", %8, %9\n"
#define F_v_pk_lshlrev_b16(d) "v_pk_lshlrev_b16 " d ", 1, " d "\n"
#define F_v_pk_ashrrev_i16(d) "v_pk_ashrrev_i16 " d ", 1, " d "\n"
#define F_v_mad_i16(d) "v_mad_i16 " d ", " d ", %8, %9\n"
#define F_v_mad_i32_i16(d) "v_mad_i32_i16 " d ", " d ", %8, %9\n"
#define F_v_dot2_i32_i16(d) "v_dot2_i32_i16 " d ", %8, %9, " d "\n"
#define F_v_dot4_i32_i8(d) "v_dot4_i32_i8 " d ", %8, %9, " d "\n"
#define F_v_cvt_pk_u8_f32(d) "v_cvt_pk_u8_f32 " d ", %8, 1, " d "\n"
#define F_v_add_u32_dpp_row_shr(d) "v_add_u32_dpp " d ", %8, " d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_v_add_u32_dpp_row_shr_self(d) "v_add_u32_dpp " d ", " d ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_v_add_u32_dpp_row_bcast(d) "v_add_u32_dpp " d ", " d ", " d " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define F_v_mov_b32_dpp_quad(d) "v_mov_b32_dpp " d ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_v_add_u32_sdwa(d) "v_add_u32_sdwa " d ", " d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0\n"
#define F_v_readlane_b32(d) "v_readlane_b32 s20, " d ", 5\n"
#define F_v_readfirstlane_b32(d) "v_readfirstlane_b32 s20, " d "\n"
#define F_s_add_u32(d) "s_add_u32 s20, s20, 1\n"
#define F_s_nop_0(d) "s_nop 0\n"
#define F_ds_read_u16(d) "ds_read_u16 " d ", %8 offset:2\n"
#define F_ds_read_i16(d) "ds_read_i16 " d ", %8 offset:2\n"
#define F_ds_read_u8(d) "ds_read_u8 " d ", %8 offset:1\n"
#define F_ds_read_b32(d) "ds_read_b32 " d ", %8 offset:4\n"
#define F_ds_read_b64(d) "ds_read_b64 " d ", %9 offset:8\n"
#define F_ds_read2_b32(d) "ds_read2_b32 " d ", %8 offset0:1 offset1:2\n"
#define F_ds_write2_b32(d) "ds_write2_b32 %8, " d ", " d " offset0:1 offset1:2\n"
#define F_ds_write_b8(d) "ds_write_b8 %8, " d " offset:1\n"
#define F_ds_write_b64(d) "ds_write_b64 %9, " d " offset:8\n"
#define F_ds_write_b16(d) "ds_write_b16 %8, " d " offset:2\n"
#define F_ds_write_b32(d) "ds_write_b32 %8, " d " offset:4\n"
#define F_ds_bpermute_b32(d) "ds_bpermute_b32 " d ", %9, %8\n"
#define F_ds_swizzle_b32(d) "ds_swizzle_b32 " d ", %8 offset:0x041F\n"

// X(name, chain registers: a32 or a64, class: 0 = VALU (also measured as one dependent chain), 1 = VALU whose result needs
// wait states before a dependent read (DPP: eight chains only), 2 = LDS (an s_waitcnt lgkmcnt(0) after every eight; %8 = the
// lane's LDS address))
#define OPS(X)                          \
    X(v_fma_f32, a32, 0)                \
    X(v_pk_fma_f32, a64, 0)             \
    X(v_mov_b32, a32, 0)                \
    X(v_add_u32, a32, 0)                \
    X(v_sub_u32, a32, 0)                \
    X(v_add_u32_sgpr, a32, 0)           \
    X(v_add3_u32, a32, 0)               \
    X(v_lshl_add_u32, a32, 0)           \
    X(v_and_b32, a32, 0)                \
    X(v_xor_b32, a32, 0)                \
    X(v_and_or_b32, a32, 0)             \
    X(v_or3_b32, a32, 0)                \
    X(v_bfi_b32, a32, 0)                \
    X(v_bitop3_b32, a32, 0)             \
    X(v_bfe_u32, a32, 0)                \
    X(v_bfe_i32, a32, 0)                \
    X(v_lshlrev_b32, a32, 0)            \
    X(v_lshrrev_b32, a32, 0)            \
    X(v_ashrrev_i32, a32, 0)            \
    X(v_ashrrev_i32_v, a32, 0)          \
    X(v_alignbit_b32, a32, 0)           \
    X(v_alignbit_b32_v, a32, 0)         \
    X(v_perm_b32, a32, 0)               \
    X(v_perm_b32_sgpr, a32, 0)          \
    X(v_cndmask_b32, a32, 0)            \
    X(v_cndmask_b32_sgpr, a32, 0)       \
    X(v_cndmask_b32_inl, a32, 0)        \
    X(v_bitop3_b32_sgpr, a32, 0)        \
    X(v_and_b32_sgpr, a32, 0)           \
    X(v_sat_pk_u8_i16, a32, 0)          \
    X(v_cndmask_b32_e64_vcc, a32, 0)    \
    X(pair_cmp_cndmask_vcc, a32, 0)     \
    X(pair_cmp_cndmask_sgpr, a32, 0)    \
    X(pair_cmp_other_cndmask_vcc, a32, 0) \
    X(v_or_b32, a32, 0)                 \
    X(v_not_b32, a32, 0)                \
    X(v_and_b32_lit, a32, 0)            \
    X(v_add_u32_inl, a32, 0)            \
    X(v_subrev_u32, a32, 0)             \
    X(v_add_co_u32, a32, 0)             \
    X(v_lshlrev_b32_v, a32, 0)          \
    X(v_lshrrev_b32_8, a32, 0)          \
    X(v_mul_u32_u24, a32, 0)            \
    X(v_lshl_or_b32, a32, 0)            \
    X(v_add_lshl_u32, a32, 0)           \
    X(v_xad_u32, a32, 0)                \
    X(v_max3_i32, a32, 0)               \
    X(v_lshl_add_u64, a64, 0)           \
    X(v_cvt_f32_ubyte0, a32, 0)         \
    X(v_cvt_f32_i32, a32, 0)            \
    X(v_rcp_f32, a32, 0)                \
    X(v_mul_f32, a32, 0)                \
    X(v_add_f32, a32, 0)                \
    X(v_mac_f32, a32, 0)                \
    X(v_pk_add_f16, a32, 0)             \
    X(v_pk_fma_f16, a32, 0)             \
    X(v_cmp_lt_i32, a32, 0)             \
    X(v_cmp_lt_i32_sdst, a32, 0)        \
    X(v_min_i32, a32, 0)                \
    X(v_max_i32, a32, 0)                \
    X(v_med3_i32, a32, 0)               \
    X(v_mul_i32_i24, a32, 0)            \
    X(v_mul_i32_i24_sgpr, a32, 0)       \
    X(v_mad_i32_i24, a32, 0)            \
    X(v_mad_i32_i24_sgpr, a32, 0)       \
    X(v_mad_u32_u24, a32, 0)            \
    X(v_mul_lo_u32, a32, 0)             \
    X(v_mul_hi_u32, a32, 0)             \
    X(v_mul_hi_i32_i24, a32, 0)         \
    X(v_mad_u64_u32, a64, 0)            \
    X(v_lerp_u8, a32, 0)                \
    X(v_sad_u8, a32, 0)                 \
    X(v_pk_add_u16, a32, 0)             \
    X(v_pk_add_i16, a32, 0)             \
    X(v_pk_sub_i16, a32, 0)             \
    X(v_pk_min_i16, a32, 0)             \
    X(v_pk_max_i16, a32, 0)             \
    X(v_pk_mul_lo_u16, a32, 0)          \
    X(v_pk_mad_i16, a32, 0)             \
    X(v_pk_lshlrev_b16, a32, 0)         \
    X(v_pk_ashrrev_i16, a32, 0)         \
    X(v_mad_i16, a32, 0)                \
    X(v_mad_i32_i16, a32, 0)            \
    X(v_dot2_i32_i16, a32, 0)           \
    X(v_dot4_i32_i8, a32, 0)            \
    X(v_cvt_pk_u8_f32, a32, 0)          \
    X(v_add_u32_dpp_row_shr, a32, 1)    \
    X(v_add_u32_dpp_row_shr_self, a32, 1) \
    X(v_add_u32_dpp_row_bcast, a32, 1)  \
    X(v_mov_b32_dpp_quad, a32, 1)       \
    X(v_add_u32_sdwa, a32, 0)           \
    X(v_readlane_b32, a32, 1)           \
    X(v_readfirstlane_b32, a32, 1)      \
    X(s_add_u32, a32, 1)                \
    X(s_nop_0, a32, 1)                  \
    X(ds_read_u8, a32, 2)               \
    X(ds_read_u16, a32, 2)              \
    X(ds_read_i16, a32, 2)              \
    X(ds_read_b32, a32, 2)              \
    X(ds_read_b64, a64, 2)              \
    X(ds_read2_b32, a64, 2)             \
    X(ds_write2_b32, a32, 2)            \
    X(ds_write_b8, a32, 2)              \
    X(ds_write_b64, a64, 2)             \
    X(ds_write_b16, a32, 2)             \
    X(ds_write_b32, a32, 2)             \
    X(ds_bpermute_b32, a32, 2)          \
    X(ds_swizzle_b32, a32, 2)

enum OpId {
#define X(n, r, c) OP_##n,
    OPS(X)
#undef X
        OP_COUNT
};
static const char* kOpNames[] = {
#define X(n, r, c) #n,
    OPS(X)
#undef X
};
static const int kOpClass[] = {
#define X(n, r, c) c,
    OPS(X)
#undef X
};

typedef unsigned long long u64;

#define I8(F) F("%0") F("%1") F("%2") F("%3") F("%4") F("%5") F("%6") F("%7")
#define I1(F) F("%0") F("%0") F("%0") F("%0") F("%0") F("%0") F("%0") F("%0")
#define R16(x) x x x x x x x x x x x x x x x x

template <int kOp, int kChains>
__global__ __launch_bounds__(256) void k_rate(int iters, Stamp* __restrict__ stamps, unsigned* __restrict__ sink)
{
    __shared__ unsigned lds[256 * 9];
    const unsigned tid = threadIdx.x;
    for (int i = 0; i < 9; i++)
        lds[tid * 9 + i] = tid * 2654435761u + i;
    __syncthreads();
    unsigned a32[8];
    u64 a64[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a32[i] = tid * 97u + i * 13u + 5u;
        a64[i] = (u64)a32[i] << 20 | i;
    }
    unsigned b = tid | 0x01020304u, c = (tid * 3u) & 0x07060504u;
    u64 wb = 0x3f8000003f800000ull;
    const unsigned k = (unsigned)iters | 3u;  // (lands in an SGPR)
    u64 t0, r0, t1, r1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if constexpr (false) {
        }
#define X(n, regs, cls)                                                                                                              \
    else if constexpr (kOp == OP_##n)                                                                                                \
    {                                                                                                                                \
        if constexpr (cls == 2) {                                                                                                    \
            const unsigned laddr = (tid & 63) * 36u; /* 9-dword pitch: conflict-free */                                              \
            asm volatile(R16(I8(F_##n) "s_waitcnt lgkmcnt(0)\n")                                                                     \
                         : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]), "+v"(regs[3]), "+v"(regs[4]), "+v"(regs[5]), "+v"(regs[6]),  \
                           "+v"(regs[7])                                                                                             \
                         : "v"(laddr), "v"((tid & 63) * 8u), "s"(k), "v"(wb)                                                         \
                         : "memory");                                                                                                \
        } else if constexpr (kChains == 8) {                                                                                         \
            asm volatile("v_cmp_lt_u32 vcc, %8, %9\n s_mov_b64 s[22:23], 0x3333\n s_nop 4\n" R16(I8(F_##n))                                      \
                         : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]), "+v"(regs[3]), "+v"(regs[4]), "+v"(regs[5]), "+v"(regs[6]),  \
                           "+v"(regs[7])                                                                                             \
                         : "v"(b), "v"(c), "s"(k), "v"(wb)                                                                           \
                         : "vcc", "s20", "s21", "s22", "s23", "scc");                                                                              \
        } else {                                                                                                                     \
            asm volatile("v_cmp_lt_u32 vcc, %8, %9\n s_mov_b64 s[22:23], 0x3333\n s_nop 4\n" R16(I1(F_##n))                                      \
                         : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]), "+v"(regs[3]), "+v"(regs[4]), "+v"(regs[5]), "+v"(regs[6]),  \
                           "+v"(regs[7])                                                                                             \
                         : "v"(b), "v"(c), "s"(k), "v"(wb)                                                                           \
                         : "vcc", "s20", "s21", "s22", "s23", "scc");                                                                              \
        }                                                                                                                            \
    }
        OPS(X)
#undef X
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    unsigned x = lds[tid];
#pragma unroll
    for (int i = 0; i < 8; i++)
        x ^= a32[i] ^ (unsigned)a64[i] ^ (unsigned)(a64[i] >> 32);
    if (x == 0x12345679u)
        sink[0] = x;
    if ((tid & 63) == 0) {
        const unsigned wave = blockIdx.x * 4 + tid / 64;
        stamps[wave].cyc = t1 - t0;
        stamps[wave].real = r1 - r0;
        stamps[wave].begin = r0;
        stamps[wave].end = r1;
    }
}

struct Row {
    std::string name;
    int chains, w;
    double wave_ns, wave_cyc, mhz, ev_ns, span_ns;
};

template <int kOp, int kChains>
static Row run(int w, int iters, Stamp* d_st, unsigned* d_sink, hipEvent_t e0, hipEvent_t e1)
{
    const int grid = 256 * w;
    std::vector<Stamp> st(grid * 4);
    hipLaunchKernelGGL((k_rate<kOp, kChains>), dim3(grid), dim3(256), 0, 0, std::max(1, iters / 8), d_st, d_sink);  // warm up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<kOp, kChains>), dim3(grid), dim3(256), 0, 0, iters, d_st, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    // the median wave's life (what ONE wave is issued at), the clock the s_memtime counter ran at, and the span from the
    // first wave's start to the last wave's end (100 MHz counter) -- the throughput figure: the oldest wave of a SIMD wins
    // the arbitration, so the waves of a SIMD do not live equally long and a wave's own time says nothing about the SIMD's rate
    std::vector<u64> real, cyc;
    u64 b0 = ~0ull, e1x = 0;
    for (auto& s : st) {
        real.push_back(s.real);
        cyc.push_back(s.cyc);
        b0 = std::min(b0, s.begin);
        e1x = std::max(e1x, s.end);
    }
    std::sort(real.begin(), real.end());
    std::sort(cyc.begin(), cyc.end());
    const double med_real = (double)real[real.size() / 2], med_cyc = (double)cyc[cyc.size() / 2];
    const double n_instr = (double)iters * 128;
    Row r;
    r.name = kOpNames[kOp];
    r.chains = kChains;
    r.w = w;
    r.wave_ns = med_real * 10.0 / n_instr;  // 100 MHz ticks
    r.wave_cyc = med_cyc / n_instr;
    r.mhz = med_cyc / (med_real * 10.0) * 1000.0;
    r.ev_ns = ms * 1e6 / (n_instr * w);
    r.span_ns = (double)(e1x - b0) * 10.0 / (n_instr * w);
    return r;
}

template <int kOp>
static void sweep(std::vector<Row>& rows, int iters, Stamp* d_st, unsigned* d_sink, hipEvent_t e0, hipEvent_t e1, const char* only)
{
    if (only && !strstr(kOpNames[kOp], only))
        return;
    for (int w : {1, 2, 4, 8})
        rows.push_back(run<kOp, 8>(w, iters, d_st, d_sink, e0, e1));
    if (kOpClass[kOp] == 0)
        rows.push_back(run<kOp, 1>(1, iters, d_st, d_sink, e0, e1));
}

template <int kOp>
static void sweep_all(std::vector<Row>& rows, int iters, Stamp* d_st, unsigned* d_sink, hipEvent_t e0, hipEvent_t e1, const char* only)
{
    sweep<kOp>(rows, iters, d_st, d_sink, e0, e1, only);
    if constexpr (kOp + 1 < OP_COUNT)
        sweep_all<kOp + 1>(rows, iters, d_st, d_sink, e0, e1, only);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const char* only = argc > 2 ? argv[2] : nullptr;
    Stamp* d_st;
    unsigned* d_sink;
    CK(hipMalloc(&d_st, 256 * 8 * 4 * sizeof(Stamp)));  // (256 x W workgroups, four waves each)
    CK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs, clock attribute %d kHz, %d trips x 128 instructions per wave\n", prop.gcnArchName, prop.multiProcessorCount, clk, iters);
    std::vector<Row> rows;
    sweep_all<0>(rows, iters, d_st, d_sink, e0, e1, only);
    // yardstick: v_fma_f32 at the same W
    double fma[16] = {0};
    for (auto& r : rows)
        if (r.name == "v_fma_f32" && r.chains == 8)
            fma[r.w] = r.span_ns;
    printf("# per row: W = waves per SIMD (workgroups of 256 threads per CU); `issue` = (first wave's start .. last wave's end) / (W x\n"
           "# instructions per wave) = what the SIMD spends per wave64 instruction, in ns and in cycles of the s_memtime counter (whose rate\n"
           "# in that run is the MHz column); `event` the same from HIP events around the launch; `wave` = the median wave's own time per\n"
           "# instruction (its issue interval); chains = independent dependency chains inside a wave\n");
    printf("%-30s %6s %2s %9s %9s %8s %9s %9s %9s %7s\n", "opcode", "chains", "W", "issue ns", "issue cyc", "MHz", "event ns", "wave ns", "wave cyc", "x fma");
    for (auto& r : rows)
        printf("%-30s %6d %2d %9.3f %9.2f %8.0f %9.3f %9.3f %9.2f %7.2f\n", r.name.c_str(), r.chains, r.w, r.span_ns, r.span_ns * r.mhz / 1000.0,
               r.mhz, r.ev_ns, r.wave_ns, r.wave_cyc, fma[r.w] > 0 && r.chains == 8 ? r.span_ns / fma[r.w] : 0.0);
    return 0;
}
