// k_recon.hip -- macroblock reconstruction: IDCT + half-pel motion compensation + store (gfx950).
//
// ONE LANE PER 8x8 BLOCK, one launch per picture index (a P picture needs the previous picture of
// its own stream; streams, macroblocks and blocks are independent).
// Restates idct() (reference src/player.cpp:922-996), predict()/predict_zero()/mocomp()
// (732-889) and copy_block/add_block[_dc] (1151-1236):
//
//   * a lane dequantises its block's compact coefficient entries (k_parse) into a private
//     64 x int16 LDS block, reads them back into 64 registers (scaled by the pre-multipliers) and
//     runs 8 column + 8 row butterflies there -- the same scaled integer AAN arithmetic as the
//     reference;
//   * for predicted blocks nine 12-byte reference rows are fetched with aligned dword loads (the
//     reference's _src_align staging, player.cpp:739-759) and the four half-pel cases are one
//     branch-free packed-byte expression;
//   * the lane adds its 64 residuals to its 64 predicted pixels, clamps to 0..248 (PIN,
//     player.cpp:183-236) and issues eight 8-byte stores into the 16-line strip layout
//     (Frame, src/video.h:36-44).
//
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"
#include "parse_tm.h"
#include "efx_probe.h"

namespace efx {

namespace {


// v_mul_i32_i24 / v_mad_i32_i24 by name: `__mul24` is library code ((x << 8 >> 8) * (y << 8 >> 8)) that the compiler turns
// into the 24-bit multiply only where it can PROVE the operand fits -- the column pass; in the row pass it kept the two
// shifts and a quarter-rate v_mul_lo_u32 per product (40 of them: a tenth of the kernel's issue time).
__device__ inline int mul24(int a, int k)
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a));
    return r;
}
__device__ inline int mad24(int a, int k, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
}

// One 8-point pass of the reference's scaled integer IDCT (player.cpp:938-995): the same products
// and the same rounding points "(x + 128) >> 8" (integer sums are associative, so only those matter for the bits),
// written as an even / odd decomposition.
// The products use the full-rate 24-bit multiplier: every multiplied operand is a combination of AC
// coefficients only -- the DC term v0 enters through dc_sum / dc_dif and is never multiplied --
// and AC coefficients are clamped to +-2048 and scaled by a premultiplier <= 62, which bounds the
// operands by 2^19 in the column pass and 2^22.5 in the row pass (L1 norm of the linear map).  The
// low 32 bits of the 48-bit product equal the reference's wrapped 32-bit product.
// (the butterfly from its first stage on: the sums and differences of the input pairs (0,4) (2,6) (1,7) (3,5))
__device__ inline void idct8_paired(int dc_sum, int dc_dif, int c_sum, int c_dif, int p17, int m17, int p35, int m53, int& v0, int& v1,
                                    int& v2, int& v3, int& v4, int& v5, int& v6, int& v7, int half)
{
    const int c_rot = (mad24(c_dif, 362, half) >> 8) - c_sum;
    const int even0 = dc_sum + c_sum, even3 = dc_sum - c_sum;
    const int even1 = dc_dif + c_rot, even2 = dc_dif - c_rot;
    const int odd_all = p17 + p35;
    const int odd_a = (mad24(m17, 473, mad24(m53, -196, half)) >> 8) - odd_all;
    const int odd_b = odd_a - (mad24(p17 - p35, 362, half) >> 8);
    const int odd_c = -odd_b - (mad24(m53, 473, mad24(m17, 196, half)) >> 8);
    // output butterflies
    v0 = even0 + odd_all;
    v7 = even0 - odd_all;
    v1 = even1 + odd_a;
    v6 = even1 - odd_a;
    v2 = even2 - odd_b;
    v5 = even2 + odd_b;
    v3 = even3 - odd_c;
    v4 = even3 + odd_c;
}
__device__ inline void idct8(int& v0, int& v1, int& v2, int& v3, int& v4, int& v5, int& v6, int& v7, int half)
{
    // even half: inputs 0, 4 as sum / difference; inputs 2, 6 through one rotation (362 / 256);
    // odd half: inputs 1, 7 and 3, 5 as sums and differences, three rotations (473, 196, 362 over 256)
    idct8_paired(v0 + v4, v0 - v4, v2 + v6, v2 - v6, v1 + v7, v1 - v7, v3 + v5, v5 - v3, v0, v1, v2, v3, v4, v5, v6, v7, half);
}

// v_bitop3_b32: any function of three words bit by bit, table index = a << 2 | b << 1 | c (a = 0xF0, b = 0xCC, c = 0xAA).
// Full rate on gfx950 (2 cycles per wave64) where v_bfi_b32 / v_and_or_b32 / v_cndmask_b32 take 4 (profiles/r6_valu_rates.md);
// the compiler still selects those, hence by name.
template <int kTable>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    // (the builtin: behind an inline asm statement the hazard recognizer puts an `s_nop`)
    return __builtin_amdgcn_bitop3_b32(a, b, c, kTable);
}

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// byte offset of plane row inside a frame (Frame::get_y/get_cr/get_cb, player.cpp:33-46)
__device__ inline int luma_row_off(int y) { return (y >> 4) * kStripBytes + (y & 15) * kStride; }
__device__ inline int chroma_row_off(int plane, int c)
{
    return (c >> 3) * kStripBytes + ((c & 7) + (plane == 2 ? 8 : 0)) * kStride + EFX_FRAME_WIDTH;
}

__device__ inline uint32_t avg_up(uint32_t a, uint32_t b)  // per byte (a + b + 1) >> 1
{
    return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu);
}

__device__ inline uint32_t avg4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)  // per byte (a+b+c+d+2) >> 2
{
    const uint32_t m = 0x00FF00FFu;
    uint32_t e = (a & m) + (b & m) + (c & m) + (d & m) + 0x00020002u;
    uint32_t o = ((a >> 8) & m) + ((b >> 8) & m) + ((c >> 8) & m) + ((d >> 8) & m) + 0x00020002u;
    return ((e >> 2) & m) | (((o >> 2) & m) << 8);
}

// per byte (a + b + c + d + 2) >> 2 from byte-wise averages (v_lerp_u8: (x + y + (z & 1)) >> 1):
// with h1 = (a + b) >> 1, h2 = (c + d) >> 1 and l = the carry bits both halves dropped,
// the result is (h1 + h2 + 1 + l) >> 1 = ceil_avg(h1, h2) + (l & ~(h1 ^ h2) & 1).
__device__ inline uint32_t avg4_lerp(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t h1 = __builtin_amdgcn_lerp(a, b, 0), h2 = __builtin_amdgcn_lerp(c, d, 0);
    const uint32_t l = (a ^ b) & (c ^ d);
    return __builtin_amdgcn_lerp(h1, h2, 0x01010101u) + (l & ~(h1 ^ h2) & 0x01010101u);
}

}  // namespace

// IDCT pre-multiplier of raster position (r, c): round(32 s_r s_c), s_0 = 1, s_k = sqrt(2) cos(k pi / 16)
// -- the reference's scale_dct_q (player.cpp:161-170), as efx_tables.cpp builds it for the scan
// table; here as compile-time constants so that every register of the in-register IDCT is scaled
// by an immediate.
__device__ constexpr double kAan[8] = {1.0,          1.3870398453221475, 1.3065629648763766, 1.1758756024193588,
                                       1.0,          0.7856949583871022, 0.5411961001461970, 0.2758993792829431};
__device__ constexpr int premul_at(int r, int c) { return (int)(32.0 * kAan[r] * kAan[c] + 0.5); }

constexpr int kBlocksPerPicture = kMbCount * 6;
constexpr int kGroupsPerPictureK = (kBlocksPerPicture + 63) / 64;  // 25 groups of 64 blocks
constexpr int kLaneDwords = 33, kLaneData = 32;  // per-lane LDS block: 64 int16 + one dword (see k_recon)
constexpr int kLaneHalfwords = 2 * kLaneDwords;

// One lane per 8x8 block, 64 consecutive blocks of a picture (in plane-row order, see below) per wave.
//
// The previous mapping (one wave per macroblock, 48 of 64 lanes in the butterflies, transposition
// through LDS) was bound by VALU issue.  Here a lane owns a whole block: it dequantises its own
// coefficient entries into a private 64 x int16 LDS block (the only use of LDS: a register file
// cannot be indexed per lane), reads the block into 64 registers, runs the 16 butterflies of the
// 2-D IDCT in registers, forms the prediction of its 8 x 8 pixels from nine 12-byte row fetches and
// stores eight 8-byte rows.  No lane ever waits for another: no barrier, no shuffle, no scalar
// bookkeeping, every lane busy.
//
// kShared = the frames this wave reads were written, and the frames it writes will be read, by OTHER workgroups of the
// SAME launch (k_recon_all): every frame access is then a write-through / L1-bypassing `sc1` buffer access (a CU's vector
// L1 is never refreshed by another CU's stores and the XCDs' L2s are not coherent with each other:
// MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"); the hand-over itself is the
// per-stream counter of k_recon_all.
#ifndef EFX_RECON_PER_ROUND
#define EFX_RECON_PER_ROUND 3  // coefficient entries a lane takes per round of the wave's entry dealing
#endif
#ifndef EFX_RECON_LDS_PAD
#define EFX_RECON_LDS_PAD 0  // (counter experiments: unused dwords of LDS per wave, to hold k_recon to fewer waves per CU)
#endif
#ifndef EFX_RECON_STORE
#define EFX_RECON_STORE 0  // k_recon (one launch per picture index): plain 8-byte row stores
#endif
#ifndef EFX_RA_LOAD
#define EFX_RA_LOAD 1      // k_recon_all: sc1 loads of the reference frame
#endif
#ifndef EFX_RA_STORE
#define EFX_RA_STORE 2     // k_recon_all: sc1 16-byte stores by lane pairs
#endif
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
constexpr int kAuxSc1 = 16;  // cache policy of the raw buffer builtins on gfx950: sc1

// which block of the picture lane `lane` of block group `g` owns: macroblock, block inside it
struct BlockAt {
    bool have;
    int blk, mb;
};
__device__ __forceinline__ BlockAt block_at(int g, int lane)
{
    const int b_raw = g * 64 + lane;
    BlockAt o;
    o.have = b_raw < kBlocksPerPicture;
    const int b = o.have ? b_raw : kBlocksPerPicture - 1;
    // Block order inside a picture: by macroblock row, and inside it by plane row -- the 44 upper luma
    // blocks, the 44 lower ones, the 22 "cr" and the 22 "cb" blocks -- so that the lanes of a wave
    // write (and mostly read) long contiguous runs of every frame row they touch.
#if defined(EFX_RECON_MB_MAJOR)
    o.mb = b / 6;
    o.blk = b - o.mb * 6;
    return o;
#endif
    const int mbrow = b / 132, rr = b - mbrow * 132;
    int mbx;
    if (rr < 88) {
        const int h = rr >= 44, r2 = rr - 44 * h;
        o.blk = 2 * h + (r2 & 1);
        mbx = r2 >> 1;
    } else {
        const int h = rr >= 110;
        o.blk = 4 + h;
        mbx = rr - 88 - 22 * h;
    }
    o.mb = mbrow * kMbW + mbx;
    return o;
}

// kLoad: 0 = plain loads of the reference frame, 1 = sc1 buffer loads (bypass this CU's L1).
// kStore: 0 = plain 8-byte row stores, 1 = the same as sc1 (write-through) stores, 2 = sc1 16-byte stores: the lanes of a wave
// come in pairs that own horizontally adjacent blocks (even lane: the left / even one), so the pair's two 8-byte row segments are
// one aligned 16-byte unit -- the even lane stores rows 0-3 of both blocks, the odd lane rows 4-7 (a write-through store is one
// fabric write whatever its size: per byte an 8-byte one costs 2.7 x a 16-byte one, MI355X_MICROARCH.md), 3 = plain 16-byte.
template <int kLoad, int kStore, class PreStore>
__device__ __forceinline__ void recon_group(uint32_t* __restrict__ lds, const uint32_t* __restrict__ coefs,
                                            const uint32_t* __restrict__ qt_custom, uint8_t* __restrict__ frames, int ring_depth,
                                            int pic, int pos0, int first_pts, int epoch, int s, int g, const uint4 rw,
                                            PreStore&& pre_store
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL == 2
                                            , const uint32_t (&efx_abl_early)[3]
#endif
                                            EFX_PROBE_PARAM)
{
    int16_t* const cfh = reinterpret_cast<int16_t*>(lds);
    uint16_t* const s_pre = reinterpret_cast<uint16_t*>(lds + 64 * kLaneDwords);  // entries before the block in the wave
    uint8_t* const s_zd = reinterpret_cast<uint8_t*>(lds + 64 * kLaneDwords + 32);  // bit 7: an entry sits at scan position 0;
                                                                                    // bits 0-5: intra DC value >> 16
    const int lane = threadIdx.x;
    const BlockAt at = block_at(g, lane);
    const bool have = at.have;
    const int blk = at.blk, mb = at.mb;
    // ring position of this picture (k_advance): the reference's _current / _reference alternation, player.cpp:692-702
    const uint32_t q = (uint32_t)pos0 + (uint32_t)(first_pts < 0 ? pic + 1 : max(0, pic - first_pts));
    const uint32_t cur_slot = q % (uint32_t)ring_depth, ref_slot = (q - 1) % (uint32_t)ring_depth;
    uint8_t* const ring = frames + (size_t)s * ring_depth * kFrameBytes;  // the stream's frames (wave-uniform)
    uint8_t* cur = ring + (size_t)cur_slot * kFrameBytes;
    const uint8_t* ref = ring + (size_t)ref_slot * kFrameBytes;
    // (kShared) the stream's ring as a buffer: offsets are 32-bit, the cache policy rides on the instruction.  The range
    // includes the slack behind the last frame that the window rows may over-read (the pool ends with it).
    const __amdgpu_buffer_rsrc_t ring_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(ring, 0, (int)(ring_depth * kFrameBytes + 8192), 0x00020000);
    const uint32_t cur_off = cur_slot * (uint32_t)kFrameBytes, ref_off = ref_slot * (uint32_t)kFrameBytes;

    const uint32_t w_base = rw.x, w_cnt = rw.y, w_misc = rw.z, w_mv = rw.w;
    EFX_PROBE_STAMP_AFTER(3, w_base);  // the record has arrived
    // a macroblock no slice covers (or an absent picture) keeps the slot's content
    const bool live = have && (w_misc >> 24) == (uint32_t)(epoch & 0xFF);
    const uint32_t flags = (w_misc >> 16) & 0xFF;
    const bool intra = flags & 1;
    const uint64_t cnts = ((uint64_t)(w_misc & 0xFFFF) << 32) | w_cnt;  // six byte counts
    const int my_cnt = live ? (int)((cnts >> (blk * 8)) & 0xFF) : 0;
    // entries of earlier blocks of the macroblock: sum of the lower `blk` bytes
    uint32_t before = 0;
#pragma unroll
    for (int k = 0; k < 5; k++)
        before += (k < blk) ? (uint32_t)((cnts >> (k * 8)) & 0xFF) : 0u;
    const uint32_t my_base = w_base + before;

    // ---- geometry of this block's plane (predict(), player.cpp:870-889) -----------------------------
    const int mb_y = mb / kMbW, mb_x = mb - mb_y * kMbW;
    const int X = (mb_x << 5) + (int16_t)(w_mv & 0xFFFF), Y = (mb_y << 5) + (int16_t)(w_mv >> 16);
    const bool luma = blk < 4;
    // half-pel position of the block's first pixel in its plane; chroma uses the floor of the halved POSITION
    const int PX = luma ? X + (blk & 1) * 16 : (X >> 1), PY = luma ? Y + (blk >> 1) * 16 : (Y >> 1);
    const int px0 = PX >> 1, py0 = PY >> 1, hx = PX & 1, hy = PY & 1;
    const int pw = luma ? EFX_FRAME_WIDTH : EFX_FRAME_WIDTH / 2, ph = luma ? EFX_FRAME_HEIGHT : EFX_FRAME_HEIGHT / 2;
    const bool inside = px0 >= 0 && py0 >= 0 && px0 + 8 + hx <= pw && py0 + 8 + hy <= ph;
    // byte offset of plane row y inside a frame: luma rows are linear; chroma row c of "cr" / "cb"
    // sits in frame row (c >> 3) * 16 + (c & 7) [+ 8] at byte 352 (Frame, video.h:36-44)
    const int chroma_base = (blk == 5 ? 8 : 0) * kStride + EFX_FRAME_WIDTH;
    auto row_off = [&](int y) { return luma ? y * kStride : (((y >> 3) << 4) + (y & 7)) * kStride + chroma_base; };

    // ---- issue the reference fetches first: nine rows of 12 bytes from a 4-byte aligned address ------
    // (the reference's _src_align copy, player.cpp:739-759).  Bytes / rows beyond the 9 x 9 window
    // may fall outside the frame; they are never used (the frame pool ends with slack).
    uint32_t wa[9], wb[9], wc[9];
    const bool want = live && !intra;
    int16_t* mine = cfh + lane * kLaneHalfwords;
    if (!__any(want && !inside)) {
        // One byte offset for the window's first row (a 24-bit multiply: rows < 2^8), then a constant step per row; a
        // chroma window crosses from one strip's eight rows into the next strip's once, at row `jump`.  Every row is
        // loaded by every lane of a wave that predicts at all (a lane that does not -- intra, not live -- reads its
        // own position: no exec-mask region per row, and the ninth row costs less than asking whether it is needed).
        const int xa = px0 & ~3;
        const int y0 = want ? py0 : 0;
        const uint32_t off0 = (uint32_t)(luma ? mul24(y0, kStride) : mul24(((y0 >> 3) << 4) + (y0 & 7), kStride) + chroma_base) +
                              (uint32_t)(want ? xa : 0);
        const uint32_t jump = luma ? 99u : 8u - ((uint32_t)y0 & 7u);
#pragma unroll
        for (int r = 0; r < 9; r++)
            wa[r] = wb[r] = wc[r] = 0;
        if (__any(want)) {
#pragma unroll
            for (int r = 0; r < 9; r++) {
                const uint32_t off = off0 + (uint32_t)(r * kStride) + ((uint32_t)r >= jump ? 8u * kStride : 0u);
                if constexpr (kLoad == 1) {
                    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(ring_rsrc, (int)(ref_off + off), 0, kAuxSc1);
                    wa[r] = v.x;
                    wb[r] = v.y;
                    wc[r] = v.z;
                } else {
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(ref + off);
                    wa[r] = p[0];
                    wb[r] = p[1];
                    wc[r] = p[2];
                }
            }
        }
    } else {
        // some vector of this wave points outside the picture (undefined in the reference): every lane
        // builds its window pixel by pixel with clamped coordinates -- identical to the direct fetch
        // for the windows that are inside -- through its private LDS block (not yet in use)
        uint32_t* w32 = reinterpret_cast<uint32_t*>(mine);
#pragma unroll 1
        for (int i = 0; i < 27; i++) {
            const int r = i / 3, d = i - r * 3;
            uint32_t word = 0;
            if (want)
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    const int yy = clampi(py0 + r, 0, ph - 1), xx = clampi((px0 & ~3) + d * 4 + k, 0, pw - 1);
                    const uint32_t o = (uint32_t)(row_off(yy) + xx);
                    const uint32_t byte = kLoad == 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(ring_rsrc, (int)(ref_off + o), 0, kAuxSc1)
                                                     : (uint32_t)ref[o];
                    word |= (byte & 0xFF) << (k * 8);
                }
            w32[i] = word;
        }
#pragma unroll
        for (int r = 0; r < 9; r++) {
            wa[r] = w32[r * 3];
            wb[r] = w32[r * 3 + 1];
            wc[r] = w32[r * 3 + 2];
        }
    }

    // ---- own coefficient entries -> private LDS block (dequantised, not yet pre-multiplied) ----------
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(mine);
#pragma unroll
        for (int k = 0; k < kLaneData; k++)
            z[k] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the table written above is read below by other lanes
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // The 64 blocks of the wave hold very different numbers of entries (3 on average, a dozen at the
    // maximum), so a loop in which every lane walks its own list runs at a quarter of the lanes.
    // Instead the wave's entries are dealt out one per lane: entry i belongs to the last lane whose
    // exclusive prefix count is <= i (six-step search in the prefix array), is dequantised with that
    // lane's parameters and lands in that lane's block.
    // inclusive prefix sum over the wave in the vector pipe: inside rows of 16 lanes by DPP row shifts, then the rows' totals
    // by row broadcasts (six instructions; six ds_bpermute round trips before)
    uint32_t incl = (uint32_t)my_cnt;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);  // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);  // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);  // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);  // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t pre = incl - (uint32_t)my_cnt;  // entries before this block in the wave (<= 64 * 64)
    const uint32_t pre_flags = pre | (flags << 16);  // flags: bit0 intra, bits 2-6 quantiser_scale, bit7 custom matrices
    s_zd[lane] = 0;
    s_pre[lane] = (uint16_t)pre;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Three entries per lane per round: the owners are searched and all loads issued before the first entry
    // is used -- a wave is alive for as many memory round trips as it makes one after the other, and with
    // one load per loop trip the entries alone cost it three (-5.5 % per launch).
    constexpr int kPerRound = EFX_RECON_PER_ROUND;
    int own[kPerRound], own_next[kPerRound];
    uint32_t ent[kPerRound], ent_next[kPerRound];
    auto fetch = [&](uint32_t i0, int (&own)[kPerRound], uint32_t (&ent)[kPerRound]) {
#pragma unroll
        for (int k = 0; k < kPerRound; k++) {
            const uint32_t i = i0 + 64 * k;
            // (L2 = 2 L, the byte offset into s_pre: the probe's address is L2 + an immediate; "prefix <= i" as the sign
            // of a difference -- both are below 2^13 -- and the step merged in by a bit operation: three full-rate
            // instructions per step, where compare + select + shift + or cost seventeen cycles -- profiles/r6_valu_rates.md)
            uint32_t L2 = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t probe = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(s_pre) + L2 + 2 * step);
                int32_t diff = (int32_t)(i - probe);
                asm("" : "+v"(diff));  // (opaque: the compiler knows the ranges and would rebuild the compare + select)
                const uint32_t below = (uint32_t)(diff >> 31);  // all ones: prefix > i
                L2 = bitop3<0xF4>(L2, (uint32_t)(2 * step), below);             // L2 | (2 step & ~below)
            }
            const int L = (int)(L2 >> 1);
            const uint32_t o_pf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(L2 + L2), (int)pre_flags);
            const uint32_t o_base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(L2 + L2), (int)my_base);
            own[k] = L | (int)(o_pf >> 16 << 8);  // owner lane | its flags << 8
            // (always a load, beyond the end from entry 0: the compiler then knows how many loads are in flight and
            // the prediction below waits for the windows only)
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL >= 1
            // (ablation, timing only -- wrong pixels: the wave's entries as ONE contiguous run; 2: the first round's requested
            // before the record is waited for, efx_abl_early)
            ent[k] = coefs[(uint32_t)__builtin_amdgcn_readfirstlane((int)my_base) + (i < total ? i : 0u)];
#else
            ent[k] = coefs[i < total ? o_base + (i - (o_pf & 0xFFFF)) : 0u];
#endif
        }
    };
    auto apply = [&](uint32_t i0) {
#pragma unroll
        for (int k = 0; k < kPerRound; k++) {
            if (i0 + 64 * k >= total)
                continue;
            const int L = own[k] & 63;
            const uint32_t e = ent[k];
            const uint32_t f = (uint32_t)own[k] >> 8;
            const bool o_intra = f & 1;
            // k_parse's stream word: raw bits << 16 | value << 6 | scan position (parse_tm.h); an intra block's first word
            // is its DC value << 6
            const int n = e & 63;
            const int level = (o_intra && n == 0) ? (int)e >> 6 : tm_level(e);
            // scan/quantiser table entry: zz | premultiplier << 8 | intra q << 16 | non-intra q << 24
            uint32_t t = lds[n * kLaneDwords + kLaneData];
            if (f & 0x80)
                t = qt_custom[n];
            if (o_intra && n == 0) {
                // b[0] = dc << 8 (player.cpp:1065) does not fit the int16 block: the value is kept as it is, its low
                // half in position 0 (no other entry of an intra block lands there), the rest beside the flag
                cfh[L * kLaneHalfwords] = (int16_t)level;
                s_zd[L] = (uint8_t)(0x80 | ((level >> 16) & 0x3F));
            } else {
                if (n == 0)
                    s_zd[L] = 0x80;
                // dequantise, player.cpp:1110-1121; |2 level +- 1| <= 511, qscale <= 31, q <= 255
                const int q = o_intra ? (int)((t >> 16) & 0xFF) : (int)(t >> 24);
                int val = level << 1;
                if (!o_intra)
                    val += (val < 0) ? -1 : 1;
                val = __mul24(__mul24(val, (int)((f >> 2) & 31)), q);
                val = (val + ((val >> 31) & 15)) >> 4;  // division by 16 truncating toward zero
                if ((val & 1) == 0)
                    val -= (val > 0) ? 1 : -1;
                val = val > 2047 ? 2047 : (val < -2048 ? -2048 : val);
                cfh[L * kLaneHalfwords + ((t >> 8) & 63)] = (int16_t)val;  // (its slot in the paired layout, efx_tables.cpp)
            }
        }
    };
    fetch(lane, own, ent);  // (also when the wave has no entry at all: the same three loads in flight on every path)
#ifdef EFX_PROBE_FINE
    EFX_PROBE_STAMP(2);  // (fine probe: the owner search is through, the first round's entries are requested)
#endif
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL == 2
#pragma unroll
    for (int k = 0; k < kPerRound; k++)
        ent[k] = efx_abl_early[k];
#endif

    // ---- prediction of the 8 rows, in the shadow of the entry loads just issued (a wave's life is its chain of
    // memory round trips: record -> owner search -> entries; this arithmetic needs only the windows, which were
    // requested before) -- the 27 window registers die here, before the 64 IDCT registers come alive -----------
    // All four half-pel cases of mocomp() (player.cpp:767-820) are ONE expression, (a + b + c + d + 2) >> 2 per byte with
    // a = the pixel, b = the pixel to the right if hx else a, c = the pixel below if hy else a, d = below-right / below /
    // right / a: with equal operands it degenerates exactly to (a + b + 1) >> 1 and to a.  (Lanes of one wave carry
    // different vectors, so a branch per case would execute all four.)  Built from byte-wise averages (v_lerp_u8):
    // with H = (a + b) >> 1 per WINDOW row, X = a ^ b, and h2 / x2 those of the row below if hy (else the same row's),
    // the result is lerp(H, h2, 1) + (X & x2 & ~(H ^ h2) & 1).  A window row's P (eight pixels at the block's position),
    // Q (one pixel to the right if hx: byte permutes with per-lane selectors), H and X serve the output row it is `a` for
    // and the one it is `c` for.
    uint32_t pr_lo[8], pr_hi[8];
    {
        const uint32_t selP = 0x03020100u + 0x01010101u * (uint32_t)(px0 & 3), selQ = selP + 0x01010101u * (uint32_t)hx;
        const uint32_t hym = 0u - (uint32_t)hy;
        uint32_t keep = intra ? 0u : ~0u;  // (an intra block has no prediction: its window registers hold its own position)
        asm volatile("" : "+v"(keep));     // (a mask, not a condition: sixteen full-rate ANDs instead of sixteen selects)
        auto pick = [](uint32_t m, uint32_t set, uint32_t clear) {  // bit by bit: m ? set : clear
            return bitop3<0xCA>(m, set, clear);
        };
        auto row = [&](int r, uint32_t& Hl, uint32_t& Hh, uint32_t& Xl, uint32_t& Xh) {
            const uint32_t Pl = __builtin_amdgcn_perm(wb[r], wa[r], selP), Ph = __builtin_amdgcn_perm(wc[r], wb[r], selP);
            const uint32_t Ql = __builtin_amdgcn_perm(wb[r], wa[r], selQ), Qh = __builtin_amdgcn_perm(wc[r], wb[r], selQ);
            Hl = __builtin_amdgcn_lerp(Pl, Ql, 0);
            Hh = __builtin_amdgcn_lerp(Ph, Qh, 0);
            Xl = Pl ^ Ql;
            Xh = Ph ^ Qh;
        };
        uint32_t Hl, Hh, Xl, Xh;
        row(0, Hl, Hh, Xl, Xh);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t Hl1, Hh1, Xl1, Xh1;
            row(r + 1, Hl1, Hh1, Xl1, Xh1);
            const uint32_t h2l = pick(hym, Hl1, Hl), h2h = pick(hym, Hh1, Hh);
            const uint32_t x2l = pick(hym, Xl1, Xl), x2h = pick(hym, Xh1, Xh);
            pr_lo[r] = (__builtin_amdgcn_lerp(Hl, h2l, 0x01010101u) + (Xl & x2l & ~(Hl ^ h2l) & 0x01010101u)) & keep;
            pr_hi[r] = (__builtin_amdgcn_lerp(Hh, h2h, 0x01010101u) + (Xh & x2h & ~(Hh ^ h2h) & 0x01010101u)) & keep;
            Hl = Hl1;
            Hh = Hh1;
            Xl = Xl1;
            Xh = Xh1;
            // (one row at a time: all nine rows' permutes hoisted to the top cost 139 registers)
            asm volatile("" : "+v"(pr_lo[r]), "+v"(pr_hi[r]), "+v"(Hl), "+v"(Hh), "+v"(Xl), "+v"(Xh));
        }
    }

#ifdef EFX_PROBE_FINE
    EFX_PROBE_STAMP_AFTER(6, pr_lo[7]);  // (fine probe: windows arrived, prediction computed -- and, vmcnt(0), the first round's entries are there)
#endif
    if (total) {
        for (uint32_t i0 = lane;;) {  // (uniform trip count: i0 - lane < total)
            // the next round's owners are searched and its entries requested before this round's are applied
            const bool more = i0 - lane + 64 * kPerRound < total;
            if (more)
                fetch(i0 + 64 * kPerRound, own_next, ent_next);
            apply(i0);
            if (!more)
                break;
            i0 += 64 * kPerRound;
#pragma unroll
            for (int k = 0; k < kPerRound; k++) {
                own[k] = own_next[k];
                ent[k] = ent_next[k];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t zd = s_zd[lane];
    EFX_PROBE_STAMP_AFTER(5, zd);  // the entries have been dealt out: the IDCT starts
#ifdef EFX_RECON_SENS
    // (sensitivity builds, tools/exp/recon_sens.sh: which unit of the CU is the launch waiting for?  Dummy work of ONE kind
    // is added and the launch timed: full-rate vector instructions, half-rate ones, LDS reads, L1-resident loads)
    {
        uint32_t d0 = zd, d1 = zd + 1, d2 = zd + 2, d3 = zd + 3;
#if EFX_RECON_SENS == 1
#pragma unroll
        for (int i = 0; i < EFX_RECON_SENS_N / 4; i++)
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#elif EFX_RECON_SENS == 2
#pragma unroll
        for (int i = 0; i < EFX_RECON_SENS_N / 4; i++)
            asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#elif EFX_RECON_SENS == 3
#pragma unroll
        for (int i = 0; i < EFX_RECON_SENS_N / 4; i++) {
            d0 += lds[(lane * kLaneDwords + (d1 & 7)) & 2047];
            d1 += lds[(lane * kLaneDwords + 8 + (d2 & 7)) & 2047];
            d2 += lds[(lane * kLaneDwords + 16 + (d3 & 7)) & 2047];
            d3 += lds[(lane * kLaneDwords + 24 + (d0 & 7)) & 2047];
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        }
#elif EFX_RECON_SENS == 4
#pragma unroll
        for (int i = 0; i < EFX_RECON_SENS_N; i++) {
            d0 += __builtin_nontemporal_load(coefs + (my_base & ~63u) + lane + 64 * (i & 1));   // (this wave's own entries again: L1 / L2 hits)
            asm volatile("" : "+v"(d0));
        }
#endif
        if ((d0 ^ d1 ^ d2 ^ d3) == 0x9E3779B9u)
            s_zd[lane] = 1;
    }
#endif
    const bool zf = (zd & 0x80) != 0;  // an entry sits at scan position 0
    // intra DC value (22 bits: a slice adds at most 1584 differentials of +-255 to the predictor)
    const int dc_raw = (int)((uint32_t)(uint16_t)mine[0] | ((uint32_t)((int)(zd << 26) >> 26) << 16));

    // A block whose only coefficient sits at scan position 0 takes the reference's "n == 1"
    // shortcut (player.cpp:1133-1140): dc = b[0] >> 8 (floor), no IDCT; for intra blocks the
    // byte is replicated WITHOUT the 0..248 clamp (copy_block_dc, player.cpp:1175-1187).
    const bool dc_only = my_cnt == 1 && zf;

    // ---- 2-D IDCT in registers (idct(), player.cpp:922-996) --------------------------------------------
    // Blocks without entries run the butterflies on zeros (which stay zero).  A dc_only block has its
    // single value X at raster position 0; the butterflies turn that into X at all 64 positions
    // exactly, so clearing X's low byte makes the final (x + 128) >> 8 deliver X >> 8, the shortcut.
    int v[64];
    const int half = 128;
    if (total == 0) {
        // (wave-uniform) no block of this wave holds a coefficient -- skipped macroblocks, static background: what the
        // butterflies make of 64 zeros and the rounding constant is the rounding constant
#pragma unroll
        for (int i = 0; i < 64; i++)
            v[i] = 128;
    } else {
    const uint32_t* mine32 = reinterpret_cast<const uint32_t*>(mine);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        // Column c (round 6): FOUR dwords from the private block -- the rows in the pairs the butterfly's first stage combines,
        // (0,4) (2,6) (1,7) (3,5) (efx_tables.cpp lays the block out so) -- and the first stage INCLUDING the pre-multipliers
        // as one v_dot2_i32_i16 per sum / difference, x_a p_a +- x_b p_b with the two pre-multipliers packed in a constant:
        // eight instructions where eight multiplies and eight additions stood, and half the LDS reads.
        auto dot = [](uint32_t pair, int pa, int pb) {
            // (by name: the builtin is selected as v_dot2c_i32_i16, the accumulating two-operand form, plus a v_mov to clear
            // its accumulator -- what the fusion saves)
            int r;
            asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "s"((uint32_t)(uint16_t)(short)pa | ((uint32_t)(uint16_t)(short)pb << 16)));
            return r;
        };
        const uint32_t q04 = mine32[c * 4], q26 = mine32[c * 4 + 1], q17 = mine32[c * 4 + 2], q35 = mine32[c * 4 + 3];
        int dc_sum, dc_dif;
        if (c == 0) {
            // the DC term: an intra block's is the parser's value << 8 (wider than the block's int16), and a block whose only
            // coefficient it is has its low byte cleared (the reference's n == 1 shortcut, see above)
            int v0 = __mul24((int)(int16_t)(q04 & 0xFFFF), premul_at(0, 0));
            v0 = intra ? (dc_raw << 8) : v0;
            v0 = dc_only ? (v0 & ~0xFF) : v0;
            const int v4 = __mul24((int)q04 >> 16, premul_at(4, 0));
            dc_sum = v0 + v4;
            dc_dif = v0 - v4;
        } else {
            dc_sum = dot(q04, premul_at(0, c), premul_at(4, c));
            dc_dif = dot(q04, premul_at(0, c), -premul_at(4, c));
        }
        idct8_paired(dc_sum, dc_dif, dot(q26, premul_at(2, c), premul_at(6, c)), dot(q26, premul_at(2, c), -premul_at(6, c)),
                     dot(q17, premul_at(1, c), premul_at(7, c)), dot(q17, premul_at(1, c), -premul_at(7, c)),
                     dot(q35, premul_at(3, c), premul_at(5, c)), dot(q35, -premul_at(3, c), premul_at(5, c)),
                     v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c], half);
        // one butterfly at a time (volatile asm statements keep their order): with all sixteen in
        // flight the temporaries push the kernel past 128 registers and an occupancy step
        asm volatile("" : "+v"(v[c]), "+v"(v[8 + c]), "+v"(v[16 + c]), "+v"(v[24 + c]), "+v"(v[32 + c]), "+v"(v[40 + c]),
                     "+v"(v[48 + c]), "+v"(v[56 + c]));
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        // the final rounding "(x + 128) >> 8" (player.cpp:985-994): the first input reaches all eight outputs unscaled
        // (it is never multiplied), so the 128 is added there, once; the shift is left to the byte permutes below -- a
        // residual fits 16 bits, so bytes 1 and 2 of x + 128 ARE the shifted value
        v[r * 8] += 128;
        idct8(v[r * 8], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7], half);
        asm volatile("" : "+v"(v[r * 8]), "+v"(v[r * 8 + 1]), "+v"(v[r * 8 + 2]), "+v"(v[r * 8 + 3]), "+v"(v[r * 8 + 4]),
                     "+v"(v[r * 8 + 5]), "+v"(v[r * 8 + 6]), "+v"(v[r * 8 + 7]));
    }
    }

    // destination rows of the block (block(), player.cpp:1124-1131)
    const int dst0 = luma ? (mb_y * 16 + (blk >> 1) * 8) * kStride + mb_x * 16 + (blk & 1) * 8
                          : mb_y * kStripBytes + (blk - 4) * 8 * kStride + EFX_FRAME_WIDTH + mb_x * 8;
    // an intra block the parser abandoned is not stored (player.cpp:1106-1107)
    const bool stored = live && !(intra && my_cnt == 0);
    const bool clamped = !(intra && dc_only);

    // copy_block[_dc] / add_block[_dc], player.cpp:1151-1236, two pixels per instruction: residual pairs come out
    // of the 32-bit registers by byte permutes (which also apply the >> 8), prediction bytes are widened the same
    // way, then packed 16-bit add, max 0, min 248 (PIN, player.cpp:183-236) and one permute back to four bytes
    typedef short pk16 __attribute__((ext_vector_type(2)));
    auto pair = [](int hi, int lo) {  // (hi + 128) >> 8 : (lo + 128) >> 8 as two int16
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x06050201u));
    };
    auto widen = [](uint32_t p, bool upper) {  // bytes 0, 1 (or 2, 3) of p as two uint16
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(0u, p, upper ? 0x0C030C02u : 0x0C010C00u));
    };
    const short ceiling = my_cnt == 0 ? 255 : 248;
    const pk16 zero = {0, 0}, top = {ceiling, ceiling};
    uint32_t flat4 = (uint32_t)(v[0] >> 8);  // intra DC-only block: replicated exactly as copy_block_dc does, unclamped and
    flat4 |= flat4 << 8;                      // unmasked (player.cpp:1175-1187)
    flat4 |= flat4 << 16;
    uint32_t clamped_m = clamped ? ~0u : 0u;
    asm volatile("" : "+v"(clamped_m));
    pre_store();  // (k_recon_all: the place where the previous item's stores are known to have left)
    uint32_t out_lo[8], out_hi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t p_lo = pr_lo[r], p_hi = pr_hi[r];
        const int* q = v + r * 8;
        uint32_t w[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t pp = h ? p_hi : p_lo;
            pk16 a = pair(q[4 * h + 1], q[4 * h]) + widen(pp, false);
            pk16 b = pair(q[4 * h + 3], q[4 * h + 2]) + widen(pp, true);
            a = __builtin_elementwise_min(__builtin_elementwise_max(a, zero), top);
            b = __builtin_elementwise_min(__builtin_elementwise_max(b, zero), top);
            w[h] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x06040200u);
        }
        // prediction only (skipped macroblock, block without coefficients) / intra DC-only replica / the clamped sum
        out_lo[r] = bitop3<0xCA>(clamped_m, w[0], flat4);
        out_hi[r] = bitop3<0xCA>(clamped_m, w[1], flat4);
        if constexpr (kStore < 2) {
            if (stored) {
                if constexpr (kStore == 1) {
                    const u32x2 o2 = {out_lo[r], out_hi[r]};
                    __builtin_amdgcn_raw_buffer_store_b64(o2, ring_rsrc, (int)(cur_off + (uint32_t)(dst0 + r * kStride)), 0, kAuxSc1);
                } else
                    *reinterpret_cast<uint2*>(cur + dst0 + r * kStride) = make_uint2(out_lo[r], out_hi[r]);
            }
        }
    }
    if constexpr (kStore >= 2) {
        // lanes 2k, 2k + 1 own the left and the right half of a 16-byte aligned unit of every row they write (block_at: luma
        // blocks alternate left / right, chroma blocks run along the macroblock row and a row of them starts on an even lane)
        const bool odd = lane & 1;
        const bool partner_stored = __builtin_amdgcn_update_dpp(0, (int)stored, 0xB1, 0xF, 0xF, false) != 0;  // quad_perm [1,0,3,2]
        const bool both = stored && partner_stored;
        const uint32_t unit0 = cur_off + (uint32_t)dst0 - (odd ? 8u : 0u) + (odd ? 4u * kStride : 0u);  // first row this lane stores
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // what the partner needs from this lane: the even lane's rows 4-7, the odd lane's rows 0-3
            const uint32_t give_lo = odd ? out_lo[k] : out_lo[4 + k], give_hi = odd ? out_hi[k] : out_hi[4 + k];
            const uint32_t got_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give_lo, 0xB1, 0xF, 0xF, false);
            const uint32_t got_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give_hi, 0xB1, 0xF, 0xF, false);
            const uint32_t own_lo = odd ? out_lo[4 + k] : out_lo[k], own_hi = odd ? out_hi[4 + k] : out_hi[k];
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 unit = {odd ? got_lo : own_lo, odd ? got_hi : own_hi, odd ? own_lo : got_lo, odd ? own_hi : got_hi};
            if (both) {
                if constexpr (kStore == 2)
                    __builtin_amdgcn_raw_buffer_store_b128(unit, ring_rsrc, (int)(unit0 + (uint32_t)(k * kStride)), 0, kAuxSc1);
                else
                    *reinterpret_cast<uint4*>(ring + unit0 + (uint32_t)(k * kStride)) = make_uint4(unit.x, unit.y, unit.z, unit.w);
            }
        }
        if (__any(stored && !partner_stored)) {  // (damaged or incomplete pictures only: a block whose neighbour is not written)
            if (stored && !partner_stored) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const u32x2 o2 = {out_lo[r], out_hi[r]};
                    if constexpr (kStore == 2)
                        __builtin_amdgcn_raw_buffer_store_b64(o2, ring_rsrc, (int)(cur_off + (uint32_t)(dst0 + r * kStride)), 0, kAuxSc1);
                    else
                        *reinterpret_cast<uint2*>(cur + dst0 + r * kStride) = make_uint2(out_lo[r], out_hi[r]);
                }
            }
        }
    }
}

// ---- one launch per picture index -------------------------------------------------------------------------------------------
// grid = (streams, 25): blockIdx.x = stream, blockIdx.y = group of 64 consecutive 8x8 blocks.  With a stream count that is a
// multiple of 8 all blocks of a stream are dispatched to XCD (stream % 8).  A P picture needs the previous picture of its
// own stream: the launch boundary is the dependency.
__global__ __launch_bounds__(64) void k_recon(const MbRec* __restrict__ mbrecs, const uint32_t* __restrict__ coefs,
                                              const uint32_t* __restrict__ scan_tab,
                                              const uint32_t* __restrict__ qtab_custom, uint8_t* __restrict__ frames,
                                              int max_pictures, int ring_depth, int pic, const int32_t* __restrict__ call_pos,
                                              int epoch, int stream0)
{
    // 8640 bytes of LDS per wave (18 waves per CU; the 91 registers allow 20).  Per lane 33 dwords: 64 int16
    // coefficients and, in the 33rd (which makes the stride odd: lanes fan out over the banks), entry `lane` of the
    // default scan / quantiser table; the wave's prefix counts for the owner search; 64 bytes of per-block flags.
    // What else a lane needs to know about another lane's block (first entry, quantiser) is fetched from that lane's
    // registers (ds_bpermute).  Measured: this layout at 16 / 18 / 19 waves per CU 8.07 / 8.16 / 7.85 M frames/s
    // (19 with the search through ds_bpermute as well), the previous one (9.7 KB, 16 waves) 7.78.
    __shared__ uint32_t lds[64 * kLaneDwords + 32 + 16 + EFX_RECON_LDS_PAD];
    const int lane = threadIdx.x;
    const int s = stream0 + blockIdx.x;
    // (one record per wave, placed by launch: the ring holds ten launches of 512 streams)
    EFX_PROBE_CLAIM_AT(2, blockIdx.x * gridDim.y + blockIdx.y,
                       ((long long)(epoch & 0xFF) * 16 + pic) * (gridDim.x * gridDim.y) + blockIdx.x * gridDim.y + blockIdx.y,
                       (unsigned)(epoch & 0xFF));
    EFX_PROBE_STAMP(1);
    EFX_PROBE_CYCLES_BEGIN();
#ifndef EFX_PROBE_FINE
    EFX_PROBE_SET(6, (unsigned long long)pic | (unsigned long long)(epoch & 0xFF) << 8 | (unsigned long long)stream0 << 16);
#endif
    // scan/quantiser table entry: zz | premultiplier << 8 | intra q << 16 | non-intra q << 24
    lds[lane * kLaneDwords + kLaneData] = scan_tab[lane];
#if defined(EFX_RECON_ITEMS2)
    // (experiment, round 6: TWO block groups per wave, the second group's record requested before the first group's work -- its
    // round trip is off the chain for half the items; the grid is (streams, 13))
    {
        const int g0 = 2 * blockIdx.y, g1 = g0 + 1;
        const MbRec* recs = mbrecs + ((size_t)s * max_pictures + pic) * kMbCount;
        const uint4 rw0 = *reinterpret_cast<const uint4*>(recs + block_at(g0, lane).mb);
        const uint4 rw1 = *reinterpret_cast<const uint4*>(recs + block_at(g1 < kGroupsPerPictureK ? g1 : g0, lane).mb);
        recon_group<0, EFX_RECON_STORE>(lds, coefs, qtab_custom + ((size_t)s * max_pictures + pic) * 64, frames, ring_depth, pic, call_pos[2 * s],
                                        call_pos[2 * s + 1], epoch, s, g0, rw0, [] {} EFX_PROBE_PASS);
        if (g1 < kGroupsPerPictureK) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            recon_group<0, EFX_RECON_STORE>(lds, coefs, qtab_custom + ((size_t)s * max_pictures + pic) * 64, frames, ring_depth, pic, call_pos[2 * s],
                                            call_pos[2 * s + 1], epoch, s, g1, rw1, [] {} EFX_PROBE_PASS);
        }
        return;
    }
#endif
    const BlockAt at = block_at(blockIdx.y, lane);
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL == 2
    uint32_t efx_abl_early[3];
    {
        const uint32_t fake = (uint32_t)((((size_t)s * max_pictures + pic) * 25 + blockIdx.y) * 256u);
        for (int k = 0; k < 3; k++)
            efx_abl_early[k] = coefs[fake + lane + 64 * k];
    }
#endif
    const uint4 rw = *reinterpret_cast<const uint4*>(mbrecs + ((size_t)s * max_pictures + pic) * kMbCount + at.mb);
    EFX_PROBE_STAMP_AFTER(3, rw.x);  // the record has arrived
    recon_group<0, EFX_RECON_STORE>(lds, coefs, qtab_custom + ((size_t)s * max_pictures + pic) * 64, frames, ring_depth, pic, call_pos[2 * s],
                       call_pos[2 * s + 1], epoch, s, blockIdx.y, rw, [] {}
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL == 2
                       , efx_abl_early
#endif
                       EFX_PROBE_PASS);
    EFX_PROBE_STAMP(4);
#ifndef EFX_PROBE_FINE
    EFX_PROBE_CYCLES_END(2);
#endif
}

// ---- all pictures of a decode call in ONE launch ----------------------------------------------------------------------------
// The per-picture launches above each pay one wave life of filling and draining (a launch of waves that live 10 us costs
// its work plus about 10 us: 12 x that per call, profiles/r4_ablations.md section 2), and every wave starts with an exposed
// round trip for its macroblock record.  Here the waves are persistent: the grid is what the chip holds, and a wave pulls
// ITEMS -- (picture, stream, group of 64 blocks) -- off a counter until none is left.
//   * Order: picture-major, so the item a P picture depends on -- the same stream's previous picture, all 25 groups of it --
//     was handed out a whole picture's worth of items earlier.  What orders them is a per-stream counter of finished items:
//     a wave polls done[s] >= 25 * picture (one relaxed sc1 load, issued an item ahead; a spin only near the end of
//     small batches), and adds 1 when its stores have left (s_waitcnt vmcnt(0), then one relaxed agent-scope atomic).
//     A wave only ever waits for items handed out BEFORE its own, each of which is in the hands of a running wave: no
//     cycle, whatever the residency and whatever else shares the chip.
//   * Eight queues, one per XCD: queue x holds the streams stream0 + x + 8 j, and a wave serves the queue of the XCD it runs
//     on (HW_REG_XCC_ID) until that is empty, then helps the next one -- a stream's frames stay in one L2, and eight heads
//     take the dequeue traffic (one head saturates at ~88 per us: MI355X_MICROARCH.md, "dequeue").  Placement is for speed
//     only: every frame access is sc1 on both sides (recon_group<true>), correct wherever an item runs.
//   * A wave works ahead of itself: the next item's macroblock record and its stream's counter are requested, and the index
//     of the item after that is claimed, before the current item's first instruction -- the record's round trip (1.5 of a
//     wave's 10.6 us under load) is off the chain, and so are the dequeue and the poll.
// sync: queue heads, a spin count (diagnostics), an abort word, and per stream the count of its finished items -- EVERY word
// in its own 128-byte line: agent-scope atomics on one line take their turns at the memory side (~88 per us per line,
// MI355X_MICROARCH.md "dequeue"); with the eight heads in one line the whole launch ran at that rate, 5.7 instead of
// 0.9 ms per 307 200 items.  Zeroed by the host before every launch.  Restates, for a whole call, the order MpegDecoder::run() gives one stream: picture after picture
// (player.cpp:692-702).
// Ordering (round-5 ADVICE asked for release / acquire or the reason they are not needed): the hand-over is correct with
// RELAXED agent-scope atomics because no plain cached access is involved on either side -- every frame store of an item is a
// write-through `sc1` store and the signal goes out only after `s_waitcnt vmcnt(0)` has seen them complete (drain(); memory
// operations of a wave complete in issue order), every frame load of the dependent item is an L1-bypassing `sc1` load issued
// after the poll returned the count, and both go to the same L2 / fabric point of coherence.  A release fence would write back
// an L2 that holds no dirty frame line, an acquire would invalidate an L1 that holds no frame line (MI355X_MICROARCH.md,
// "Valid forms": {sc0 sc1 stores and loads both sides}); the plain-access build, which does need them, fails parity.
constexpr unsigned long long kReconAllPatienceTicks = 1000000000ull;  // 10 s of the 100 MHz wall clock
constexpr int kGroupsPerPicture = (kBlocksPerPicture + 63) / 64;  // 25
constexpr uint32_t kSyncLine = 32;  // words per line
constexpr uint32_t kSyncHeads = 0, kSyncSpins = 8 * kSyncLine, kSyncAbort = 9 * kSyncLine, kSyncDone = 64 * kSyncLine;
#ifdef EFX_RA_STATS
// (development build: where a wave's time goes, summed over the launch -- lines 16 ... of the header, one word per line;
// read through efx_debug_recon_stats, tools/exp/r5_recon_check.py)
constexpr uint32_t kSyncStats = 16 * kSyncLine;
enum { kStWaves, kStItems, kStForeign, kStClaim, kStDep, kStBody, kStSignal, kStLife, kStXcc0 /* .. +7 */, kStCount = kStXcc0 + 8 };
#define EFX_RA_T() wall_clock64()
#else
#define EFX_RA_T() 0ull
#endif

struct ItemAt {
    int pic, s, g;
};

__device__ __forceinline__ void recon_all_body(uint32_t* __restrict__ lds, const MbRec* __restrict__ mbrecs,
                                               const uint32_t* __restrict__ coefs, const uint32_t* __restrict__ scan_tab,
                                               const uint32_t* __restrict__ qtab_custom, uint8_t* __restrict__ frames,
                                               int max_pictures, int ring_depth, int n_pictures,
                                               const int32_t* __restrict__ call_pos, int epoch, int stream0, int n_streams,
                                               uint32_t* __restrict__ sync, uint32_t* __restrict__ status, int max_items)
{
    const int lane = threadIdx.x;
    lds[lane * kLaneDwords + kLaneData] = scan_tab[lane];  // (once per wave: the items leave the 33rd dword alone)
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // queue q: streams stream0 + q + 8 j, j < per_queue(q); its items picture by picture, inside a picture stream by stream, a
    // stream's 25 groups together: the last group of a stream's picture p and the first of its picture p + 1 are a whole
    // picture of the queue apart (with the streams innermost they were per_queue(q) items apart -- fewer than the waves in
    // flight: every picture began with a wait)
    auto per_queue = [&](uint32_t q) { return (uint32_t)(n_streams > (int)q ? (n_streams - (int)q + 7) >> 3 : 0); };
    auto items_of = [&](uint32_t q) { return per_queue(q) * (uint32_t)(kGroupsPerPicture * n_pictures); };
    uint32_t q = xcc & 7, visited = 0;
    auto decode = [&](uint32_t qq, uint32_t idx) {
        const uint32_t per_pic = per_queue(qq) * kGroupsPerPicture;
        const uint32_t p = idx / per_pic, rem = idx - p * per_pic, j = rem / kGroupsPerPicture, g = rem - j * kGroupsPerPicture;
        return ItemAt{(int)p, stream0 + (int)(qq + 8 * j), (int)g};
    };
    // a returning agent-scope add on the queue head, by lane 0; the value is waited for where it is used
    auto claim_issue = [&](uint32_t qq) {
        uint32_t v = 0;
        if (lane == 0)
            v = __hip_atomic_fetch_add(sync + kSyncHeads + qq * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    // walk the queues until one hands out an item (a synchronous claim: the start of a wave, the end of a queue); false when
    // all eight are empty
    auto claim_walk = [&](uint32_t& idx) {
        while (visited < 8) {
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim_issue(q));
            if (idx < items_of(q))
                return true;
            q = (q + 1) & 7;
            visited++;
        }
        return false;
    };
    auto record_of = [&](const ItemAt& it) {
        const BlockAt at = block_at(it.g, lane);
        return *reinterpret_cast<const uint4*>(mbrecs + ((size_t)it.s * max_pictures + it.pic) * kMbCount + at.mb);
    };
    auto done_of = [&](const ItemAt& it) {
        return __hip_atomic_load(sync + kSyncDone + (uint32_t)(it.s - stream0) * kSyncLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto signal = [&](int s_done) {  // one more finished item of the stream (the caller knows that its stores have left)
        if (lane == 0)
            __hip_atomic_fetch_add(sync + kSyncDone + (uint32_t)(s_done - stream0) * kSyncLine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto drain = [] { __builtin_amdgcn_s_waitcnt(0x0F70); };  // s_waitcnt vmcnt(0), as a builtin: the compiler's own bookkeeping
                                                               // learns that every load and atomic issued so far has returned

    // A wave takes at most `max_items` items and ends (0: as many as there are): the grid is then items / max_items
    // workgroups and wave slots keep coming free -- a grid of waves that live for the whole call would hold every CU's LDS
    // until its last item, and the parse kernel of the NEXT call (which must run beside this one, efx_api.hip) would find no
    // room.  An item that has been claimed is always processed: claims go out only while the budget has room.
    //
    // The pipeline of a wave, three items deep: A is worked on; B's macroblock record and its stream's counter were requested
    // at the top of A's turn; C's index is being claimed.  ALL of that is waited for at ONE place -- just before A's first store,
    // where every load of A has long been consumed and the wait is free -- and there the hand-over signal of the item BEFORE A
    // goes out too (memory operations complete in issue order: its stores have left).  A wait anywhere else would sit out A's
    // stores: the counter of outstanding memory operations cannot tell an old atomic from a new store.
    uint32_t budget = max_items > 0 ? (uint32_t)max_items : 0xFFFFFFFFu;
    [[maybe_unused]] const unsigned long long st_begin = EFX_RA_T();
    [[maybe_unused]] unsigned long long st_top = 0, st_dep = 0, st_body = 0;
    [[maybe_unused]] uint32_t st_items = 0, st_foreign = 0;
    uint32_t idx = 0;
    if (!claim_walk(idx))
        return;
    budget--;
    ItemAt a = decode(q, idx);
    bool have_b = budget > 0 && claim_walk(idx);
    ItemAt b = a;
    if (have_b) {
        budget--;
        b = decode(q, idx);
    }
    uint4 rw_a = record_of(a);
    uint32_t seen_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)done_of(a));
    int pending_signal = -1;  // stream of the item whose stores are still on their way
    for (;;) {
        [[maybe_unused]] const unsigned long long st_t0 = EFX_RA_T();
        // ---- requests for the items behind this one (consumed at this item's pre-store point) -----------------------------------
        uint4 rw_b = rw_a;
        uint32_t done_b = 0, claim_c = 0;
        const uint32_t q_c = q;
        const bool pending_c = have_b && budget > 0;
        if (have_b) {
            rw_b = record_of(b);
            done_b = done_of(b);
        }
        if (pending_c)
            claim_c = claim_issue(q_c);
        // ---- this item's predecessor: every group of the stream's previous pictures is in memory ---------------------------
        [[maybe_unused]] const unsigned long long st_t1 = EFX_RA_T();
        const uint32_t need = (uint32_t)(kGroupsPerPicture * a.pic);
        if (seen_a < need) {
            // (the tail of a small batch: the predecessor is still in the hands of another wave -- or of THIS one: the item
            // whose signal is still held back may be what this item waits for)
            if (pending_signal >= 0) {
                drain();
                signal(pending_signal);
                pending_signal = -1;
            }
            // The wait is bounded by WALL-CLOCK time (round-5 ADVICE: a count of polls -- 2^19 x s_sleep(8), 0.1-0.2 s -- can run
            // out under a legitimate wait on a shared or preempted GPU): kReconAllPatienceTicks of the 100 MHz counter = 10 s,
            // looked at every 1024 polls.  When it does run out the stream is flagged EFX_STREAM_INTERNAL and the launch
            // gives up: the frames of that call are NOT valid (include/efx.h).
            uint32_t spins = 0;
            bool abort = false;
            const unsigned long long t_wait0 = wall_clock64();
            do {
                __builtin_amdgcn_s_sleep(8);
                seen_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)done_of(a));
                if ((++spins & 1023) == 0)
                    abort = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(sync + kSyncAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 ||
                            wall_clock64() - t_wait0 > kReconAllPatienceTicks;
            } while (seen_a < need && !abort);
            if (seen_a < need && lane == 0) {
                // never observed: a lost hand-over must not hang the device -- the stream is flagged, every other wait of the
                // launch gives up at its next look, the call ends
                atomicOr(status + a.s, EFX_STREAM_INTERNAL);
                __hip_atomic_store(sync + kSyncAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0)
                __hip_atomic_fetch_add(sync + kSyncSpins, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        [[maybe_unused]] const unsigned long long st_t2 = EFX_RA_T();
        uint32_t seen_b = 0, idx_c = 0;
        recon_group<EFX_RA_LOAD, EFX_RA_STORE>(lds, coefs, qtab_custom + ((size_t)a.s * max_pictures + a.pic) * 64, frames, ring_depth, a.pic,
                                               call_pos[2 * a.s], call_pos[2 * a.s + 1], epoch, a.s, a.g, rw_a, [&] {
                                                   drain();
                                                   if (pending_signal >= 0)
                                                       signal(pending_signal);
                                                   seen_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)done_b);
                                                   idx_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim_c);
                                               }
#if defined(EFX_RECON_ABL) && EFX_RECON_ABL == 2
                                               , {0u, 0u, 0u}
#endif
                                               EFX_PROBE_PASS_NONE);
        pending_signal = a.s;
#ifdef EFX_RA_STATS
        {
            const unsigned long long st_t3 = EFX_RA_T();
            st_top += st_t1 - st_t0;
            st_dep += st_t2 - st_t1;
            st_body += st_t3 - st_t2;
            st_items++;
            st_foreign += (uint32_t)((a.s - stream0) & 7) != (xcc & 7);
        }
#endif
        if (!have_b)
            break;
        // ---- rotate: B becomes the item worked on, C (if its claim found one) the item behind it --------------------------
        bool have_c = false;
        ItemAt c = b;
        if (pending_c) {
            have_c = idx_c < items_of(q_c);
            if (!have_c) {  // that queue is empty: on to the next ones (synchronous, a handful of times per wave at most)
                if (q == q_c) {
                    q = (q + 1) & 7;
                    visited++;
                }
                have_c = claim_walk(idx_c);
            }
            if (have_c) {
                budget--;
                c = decode(q, idx_c);
            }
        }
        a = b;
        rw_a = rw_b;
        seen_a = seen_b;
        b = c;
        have_b = have_c;
    }
    drain();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last item's stores)
    signal(pending_signal);
#ifdef EFX_RA_STATS
    if (lane == 0) {
        auto add = [&](int k, unsigned long long v) {
            __hip_atomic_fetch_add(sync + kSyncStats + (uint32_t)k * kSyncLine, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        add(kStWaves, 1);
        add(kStItems, st_items);
        add(kStForeign, st_foreign);
        add(kStClaim, st_top >> 2);  // (100 MHz ticks / 4 = units of 40 ns)
        add(kStDep, st_dep >> 2);
        add(kStBody, st_body >> 2);
        add(kStLife, (EFX_RA_T() - st_begin) >> 2);
        add(kStXcc0 + (int)(xcc & 7), 1);
    }
#endif
}

#ifndef EFX_RECON_ALL_WAVES
#define EFX_RECON_ALL_WAVES 4  // waves per SIMD the register allocation aims at: 4 (123 registers, nothing spilled; with 5 the
                               // 96 registers spill 14 and the launch is a quarter slower) -- 16 waves per CU
#endif
__global__ __launch_bounds__(64, EFX_RECON_ALL_WAVES) void k_recon_all(
    const MbRec* __restrict__ mbrecs, const uint32_t* __restrict__ coefs, const uint32_t* __restrict__ scan_tab,
    const uint32_t* __restrict__ qtab_custom, uint8_t* __restrict__ frames, int max_pictures, int ring_depth, int n_pictures,
    const int32_t* __restrict__ call_pos, int epoch, int stream0, int n_streams, uint32_t* __restrict__ sync,
    uint32_t* __restrict__ status, int max_items)
{
    __shared__ uint32_t lds[64 * kLaneDwords + 32 + 16];
    recon_all_body(lds, mbrecs, coefs, scan_tab, qtab_custom, frames, max_pictures, ring_depth, n_pictures, call_pos, epoch, stream0,
                   n_streams, sync, status, max_items);
}

// FNV-1a-64 of whole ring frames, one lane per frame (verification helper, not on the timed path)
__global__ void k_frame_hash(const uint8_t* __restrict__ frames, int n_frames, uint64_t* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames)
        return;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(frames + (size_t)i * kFrameBytes);
    uint64_t h = 0xcbf29ce484222325ull;
    for (int k = 0; k < kFrameBytes / 4; k++) {
        uint32_t w = p[k];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            h ^= (w >> (8 * b)) & 0xFF;
            h *= 0x100000001b3ull;
        }
    }
    out[i] = h;
}

__global__ void k_fill(uint32_t* __restrict__ p, uint32_t v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        p[i] = v;
}

}  // namespace efx

EFX_PROBE_READER(recon)
