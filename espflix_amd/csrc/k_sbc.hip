// k_sbc.hip -- SBC audio frames -> PCM for a batch of streams (gfx950).
//
// Restates sbc_decoder() and everything below it (reference src/sbc_decoder.cpp:74-373:
// get_samples 276-344, bit_allocation 142-240, IQUANT 263-270, synthesize8 74-139) the way
// decode_audio() drives it (src/video.cpp:962-989): a stream is a run of equally sized frames
// decoded in order, the synthesis filter memory carrying over from frame to frame.
//
// FRAME-PARALLEL.  The synthesis filter is an FIR: the matrixing outputs of a block depend on that block's samples only, a
// PCM sample on the last ten blocks' outputs -- a workgroup takes a chunk of consecutive frames, dequantises them and the
// frames that hold the nine blocks before them, matrixes, windows.  The kernels of an efx_sbc_decode call:
//   k_sbc_frames   one thread per frame: the header's verdict and the bit allocation (SbcFrameInfo) -- the one long serial
//                  piece of a frame, with every lane of its waves busy;
//   k_sbc_plan     its last workgroup sorts the streams onto the work lists -- REGULAR (every frame decodes, one geometry;
//                  mono / stereo) or GENERAL; one wave per general stream resolves what the reference chains from frame to
//                  frame (a rejected frame is synthesised from the samples and under the geometry the state holds:
//                  sbc_decoder.cpp:346-373) into prefix scans over the frames (SbcFramePlan), and finds the granules of eight
//                  frames that are regular again (SbcExtraItem slots, cover bytes);
//   k_sbc_par_*    persistent workgroups over the (stream, chunk of 16 / 8 frames) items of the regular lists, dealt round
//                  robin, then over the general streams' slots;
//   k_sbc_gen      the same for what is left of the general streams, every index looked up in the plan instead of computed
//                  in closed form;
//   k_sbc_finish   puts the new decoder states in place (the last chunk of a stream must not overwrite the state its first
//                  chunk may still be reading) and decodes the streams no list took.
// k_sbc, one wave per stream walking its frames, stays for a state no efx_sbc_decode call can have left (a block count that
// is no multiple of four) and as the comparison the tests run the other kernels against (EFX_OPT_SBC_SERIAL).  In all of them:
//   * the reference's lazy bit reader is replaced by direct addressing -- every block of a frame
//     has the same layout, so sample (blk, ch, sb) starts at bit blk * bits_per_block +
//     prefix[ch][sb] -- one lane per sample extracts, dequantises (32-bit wrapping shift, C
//     division) and stores it;
//   * the reference's sliding buffer v[170] + 16 offsets (a 10-deep history per matrix output) is
//     kept as rows: row t holds the 16 matrixing outputs of block t.  The matrixing of ALL blocks
//     in reach is computed at once, then the windowing of all blocks reads rows t .. t-9;
//   * accumulation is 32-bit wrapping, >> 15 arithmetic, clip to +-0x7FFF exactly as the reference.
// Frames the reference rejects behave as it does: a bad sync byte re-synthesises the state's
// previous subband samples, joint stereo switches the geometry and synthesises stale samples, a
// 4-subband header yields nothing (and keeps yielding nothing until a good frame).
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

constexpr uint32_t kSbcRegularFlag = 0xFFFFFFFFu, kSbcGeneralFlag = 0, kSbcSerialFlag = 2;  // parallel[stream]
constexpr uint32_t kSbcSkipEntry = 0xFFFFFFFEu;  // SbcExtraItem::entry of a chunk that is not a regular kernel's

namespace {

__device__ inline uint32_t be_bits(const uint8_t* p, uint32_t limit, uint32_t bitpos, int n)
{
    // n <= 16 bits starting at bitpos (MSB first); bytes at or beyond `limit` read as zero
    uint32_t b = bitpos >> 3;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        w = (w << 8) | (b + k < limit ? p[b + k] : 0u);
    return (w >> (24 - (bitpos & 7) - n)) & ((1u << n) - 1);
}

// Appendix B 12.6.3 as the reference writes it (sbc_decoder.cpp:142-240), one channel
__device__ void bit_allocation(int frequency, int allocation, int bitpool, const uint8_t* scale, int* bits)
{
    const int8_t offset8[4][8] = {{-2, 0, 0, 0, 0, 0, 0, 1}, {-3, 0, 0, 0, 0, 0, 1, 2}, {-4, 0, 0, 0, 0, 0, 1, 2}, {-4, 0, 0, 0, 0, 0, 1, 2}};
    int bitneed[8];
    int max_bitneed = 0;
#pragma unroll
    for (int sb = 0; sb < 8; sb++) {
        int s = scale[sb];
        int need;
        if (allocation)
            need = s;
        else if (s == 0)
            need = -5;
        else {
            int loudness = s - offset8[frequency][sb];
            if (loudness > 0)
                loudness /= 2;
            need = loudness;
        }
        bitneed[sb] = need;
        max_bitneed = max(max_bitneed, need);
    }
    // The reference's bit-slice loop (sbc_decoder.cpp:186-198) counts, slice by slice, the subbands with bitslice + 1 < bitneed <
    // bitslice + 16 once and those with bitneed == bitslice + 1 twice.  With G(x) = subbands whose bitneed is at least x that is
    // 2 G(bitslice + 1) - G(bitslice + 2) - G(bitslice + 16), and G(bitslice + 2) is the slice before's G(bitslice + 1): two
    // counts a slice, each over the eight needs as BYTES (bitneed + 32: 27 .. 47) -- byte + (128 - x) has bit 7 set iff byte >= x.
    uint32_t e_lo = 0, e_hi = 0;
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
        e_lo |= (uint32_t)(bitneed[sb] + 32) << (8 * sb);
        e_hi |= (uint32_t)(bitneed[sb + 4] + 32) << (8 * sb);
    }
    auto at_least = [&](int need) -> int {
        const uint32_t x = (uint32_t)min(max(need + 32, 0), 128);
        const uint32_t k = __builtin_amdgcn_perm(0u, 128u - x, 0u);  // (the byte in all four)
        return __popc((e_lo + k) & 0x80808080u) + __popc((e_hi + k) & 0x80808080u);
    };
    int bitcount = 0, slicecount = 0, bitslice = max_bitneed + 1, above = 0;  // above = G(bitslice + 2) of the slice to come: none
    do {
        bitslice--;
        bitcount += slicecount;
        const int here = at_least(bitslice + 1);
        slicecount = 2 * here - above - at_least(bitslice + 16);
        above = here;
    } while (bitcount + slicecount < bitpool);
    if (bitcount + slicecount == bitpool) {
        bitcount += slicecount;
        bitslice--;
    }
#pragma unroll
    for (int sb = 0; sb < 8; sb++) {
        int b = 0;
        if (bitneed[sb] >= bitslice + 2)
            b = min(bitneed[sb] - bitslice, 16);
        bits[sb] = b;
    }
    // (both loops written out: with the subband a constant, bits[] and bitneed[] stay registers -- the rolled loops indexed
    // them through chains of selects, ten times the instructions)
#pragma unroll
    for (int sb = 0; sb < 8; sb++) {
        if (bitcount < bitpool) {
            if (bits[sb] >= 2 && bits[sb] < 16) {
                bits[sb]++;
                bitcount++;
            } else if (bitneed[sb] == bitslice + 1 && bitpool > bitcount + 1) {
                bits[sb] = 2;
                bitcount += 2;
            }
        }
    }
#pragma unroll
    for (int sb = 0; sb < 8; sb++)
        if (bitcount < bitpool && bits[sb] < 16) {
            bits[sb]++;
            bitcount++;
        }
}

}  // namespace

// grid = streams, block = 64.  frames: stream s at frames + s * stream_stride, n_frames frames of
// frame_bytes.  pcm: stream s at pcm + s * pcm_stride (int16 units), frames written back to back
// (blocks * 8 * channels samples each, channel blocks not interleaved).  ret (optional): per
// (stream, frame) the reference's return value in the low 16 bits (0xFFFF = -1) and the decoded
// byte count in the high 16.  flags bit 0: decode frame 0 once more up front and drop its PCM
// (decode_audio()'s frame-size probe, video.cpp:964-972).
__device__ __forceinline__ void sbc_serial_body(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                int n_frames, SbcState* __restrict__ states,
                                                const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                                size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                                int flags)
{
    __shared__ SbcTables tb;
    __shared__ int32_t sb_sample[16][2][8];  // the reference's sb_sample (persists across frames)
    __shared__ int32_t rows[2][9 + 16][16];   // matrixing outputs: 9 rows of history + this frame's blocks
    __shared__ int sh_bits[2][8];
    __shared__ uint8_t sh_scale[2][8];

    const int s = blockIdx.x, lane = threadIdx.x;
    SbcState* st = states + s;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tb);
        for (int i = lane; i < (int)(sizeof(SbcTables) / 4); i += 64)
            dst[i] = src[i];
        for (int i = lane; i < 256; i += 64)
            (&sb_sample[0][0][0])[i] = (&st->sb_sample[0][0][0])[i];
        for (int i = lane; i < 2 * 9 * 16; i += 64) {
            int c = i / 144, r = i - c * 144;
            rows[c][r >> 4][r & 15] = st->hist[c][r >> 4][r & 15];
        }
    }
    int frequency = st->frequency, blocks = st->blocks, channels = st->channels, mode = st->mode, allocation = st->allocation,
        subbands = st->subbands, bitpool = st->bitpool;
    // A rejected frame is synthesised "from whatever the state holds" (sbc_decoder.cpp:346-373): a state buffer
    // that was not zero-initialised must not index the LDS rows or the PCM buffer out of range.
    blocks = min(blocks, 16);
    channels = min(channels, 2);
    subbands = subbands == 4 ? 4 : 8;
    __syncthreads();

    const uint8_t* base = frames + (size_t)s * stream_stride;
    const uint32_t limit = (uint32_t)n_frames * (uint32_t)frame_bytes;
    int16_t* out = pcm + (size_t)s * pcm_stride;
    uint32_t written = 0;
    const int first = (flags & 1) && n_frames > 0 ? -1 : 0;
    for (int f = first; f < n_frames; f++) {
        const bool probe = f < 0;
        const uint32_t foff = (uint32_t)(probe ? 0 : f) * (uint32_t)frame_bytes;
        const uint8_t* d = base + foff;
        const uint32_t avail = limit - foff;  // bytes of the stream from this frame on
        int framelen = -1;
        // ---- get_samples(): header (wave-uniform) ---------------------------------------------------
        bool ok = frame_bytes >= 4 && d[0] == 0x9C;
        if (ok) {
            const uint32_t h1 = d[1];
            frequency = (h1 >> 6) & 3;
            blocks = 4 * (((h1 >> 4) & 3) + 1);
            mode = (h1 >> 2) & 3;
            channels = mode ? 2 : 1;
            allocation = (h1 >> 1) & 1;
            subbands = (h1 & 1) ? 8 : 4;
            bitpool = d[2];
            // joint stereo and 4 subbands are not decoded (sbc_decoder.cpp:293-295); a bitpool the
            // allocation loop can never meet would hang the reference: rejected the same way
            ok = mode != 3 && subbands == 8 && bitpool <= 128;
        }
        if (ok) {
            if (lane < channels * 8) {
                const uint32_t i = 4 + (lane >> 1);
                const uint32_t a = i < avail ? d[i] : 0u;
                sh_scale[lane >> 3][lane & 7] = (lane & 1) ? (a & 0xF) : (a >> 4);
            }
            __syncthreads();
            if (lane < channels) {
                int b[8];
                bit_allocation(frequency, allocation, bitpool, sh_scale[lane], b);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    sh_bits[lane][k] = b[k];
            }
            __syncthreads();
            // bit offsets inside a block: (ch, sb) order
            const int per_blk = channels * 8;
            int my_bits = 0, my_prefix = 0, per_block = 0;
            {
                int acc = 0;
                for (int c = 0; c < channels; c++)
                    for (int k = 0; k < 8; k++) {
                        if (c * 8 + k == lane % per_blk) {
                            my_prefix = acc;
                            my_bits = sh_bits[c][k];
                        }
                        acc += sh_bits[c][k];
                    }
                per_block = acc;
            }
            const uint32_t data_off = 4 + (uint32_t)(channels * 8 >> 1);
            // ---- samples: one lane per (blk, ch, sb) ---------------------------------------------------
            for (int i = lane; i < blocks * per_blk; i += 64) {
                // 64 is a multiple of per_blk (8 or 16): the lane keeps its (ch, sb) and only blk advances
                const int blk = i / per_blk, r = i - blk * per_blk;
                int32_t sample = 0;
                if (my_bits) {
                    const uint32_t bitpos = data_off * 8 + (uint32_t)blk * (uint32_t)per_block + (uint32_t)my_prefix;
                    const int scale = sh_scale[r >> 3][r & 7];
                    int32_t q = (int32_t)be_bits(d, avail, bitpos, my_bits);
                    q = (q << 1) | 1;                                                 // IQUANT, sbc_decoder.cpp:263-270
                    q = (int32_t)((uint32_t)q << scale) / ((1 << my_bits) - 1);
                    sample = q - (1 << scale);
                }
                sb_sample[blk][r >> 3][r & 7] = sample;
            }
            framelen = (int)data_off + (blocks * per_block + 7) / 8;
            __syncthreads();
        }
        // ---- sbc_decoder(): synthesis from whatever the state holds --------------------------------
        uint32_t decoded = 0;
        if (subbands != 4) {
            // matrixing, all blocks at once: rows[c][9 + blk][i] = (sum_j syn[i][j] * sb[blk][c][j]) >> 15
            for (int i = lane; i < channels * blocks * 16; i += 64) {
                const int c = i / (blocks * 16), r = i - c * blocks * 16, blk = r >> 4, o = r & 15;
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc += (uint32_t)tb.syn[o * 8 + j] * (uint32_t)sb_sample[blk][c][j];
                rows[c][9 + blk][o] = (int32_t)acc >> 15;
            }
            __syncthreads();
            // windowing: sample i of block blk from rows blk .. blk-9 (slot i, even taps) and slot i+8 (odd taps)
            for (int i = lane; i < channels * blocks * 8; i += 64) {
                const int c = i / (blocks * 8), r = i - c * blocks * 8, blk = r >> 3, o = r & 7;
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 10; j += 2) {
                    acc += (uint32_t)rows[c][9 + blk - j][o] * (uint32_t)tb.proto[o * 10 + j];
                    acc += (uint32_t)rows[c][9 + blk - j - 1][o + 8] * (uint32_t)tb.proto[o * 10 + j + 1];
                }
                int32_t v = (int32_t)acc >> 15;
                v = v < -0x7FFF ? -0x7FFF : (v > 0x7FFF ? 0x7FFF : v);
                if (!probe)
                    out[written + (uint32_t)i] = (int16_t)v;
            }
            __syncthreads();
            // keep the last nine rows as the next frame's history
            if (blocks > 0) {
                int32_t keep[5];
                int n_keep = 0;
                for (int i = lane; i < channels * 144; i += 64, n_keep++) {
                    const int c = i / 144, r = i - c * 144;
                    keep[n_keep] = rows[c][blocks + (r >> 4)][r & 15];
                }
                __syncthreads();
                n_keep = 0;
                for (int i = lane; i < channels * 144; i += 64, n_keep++) {
                    const int c = i / 144, r = i - c * 144;
                    rows[c][r >> 4][r & 15] = keep[n_keep];
                }
                __syncthreads();
            }
            decoded = (uint32_t)(blocks * subbands * channels * 2);
            if (!probe)
                written += decoded / 2;
        }
        if (ret && !probe && lane == 0)
            ret[(size_t)s * n_frames + f] = ((uint32_t)framelen & 0xFFFF) | (decoded << 16);
    }

    // ---- state out ---------------------------------------------------------------------------------
    for (int i = lane; i < 256; i += 64)
        (&st->sb_sample[0][0][0])[i] = (&sb_sample[0][0][0])[i];
    for (int i = lane; i < 2 * 9 * 16; i += 64) {
        int c = i / 144, r = i - c * 144;
        st->hist[c][r >> 4][r & 15] = rows[c][r >> 4][r & 15];
    }
    if (lane == 0) {
        st->frequency = (uint8_t)frequency;
        st->blocks = (uint8_t)blocks;
        st->channels = (uint8_t)channels;
        st->mode = (uint8_t)mode;
        st->allocation = (uint8_t)allocation;
        st->subbands = (uint8_t)subbands;
        st->bitpool = (uint8_t)bitpool;
        if (pcm_count)
            pcm_count[s] = written;
    }
}

__global__ __launch_bounds__(64) void k_sbc(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                            int n_frames, SbcState* __restrict__ states,
                                            const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                            size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                            int flags)
{
    sbc_serial_body(frames, stream_stride, frame_bytes, n_frames, states, tables, pcm, pcm_stride, ret, pcm_count, flags);
}

// (development switches: the alternatives measured in profiles/r5_sbc.md)
#ifndef EFX_SBC_M
#define EFX_SBC_M 1      // matrixing: a thread makes two outputs of 1 = two consecutive rows, 0 = one row per trip
#endif
#ifndef EFX_SBC_SKIP
#define EFX_SBC_SKIP 0   // timing ablations (wrong PCM): 1 = no dequantisation, 2 = no matrixing of the chunk's rows, 4 = no windowing
#endif
#ifndef EFX_SBC_CHUNK_MONO
#define EFX_SBC_CHUNK_MONO 16  // frames per work item of k_sbc_par_mono (8: 7 workgroups per compute unit; 16: 4, and 12 % faster)
#endif
#ifndef EFX_SBC_PITCH
#define EFX_SBC_PITCH 18 // (16 = unpadded)
#endif
#ifndef EFX_SBC_WAVES
#define EFX_SBC_WAVES 4  // k_sbc_par_mono: waves per SIMD the register allocation aims at (= workgroups per compute unit)
#endif
constexpr int kSbcChunk = 8;                 // frames per work item (k_sbc_par_mono: EFX_SBC_CHUNK_MONO)
constexpr int kSbcMonoGranules = EFX_SBC_CHUNK_MONO / kSbcChunk;  // k_sbc_gen items in a k_sbc_par_mono item
static_assert(EFX_SBC_CHUNK_MONO == kSbcChunk * kSbcMonoGranules && (kSbcMonoGranules == 1 || kSbcMonoGranules == 2), "mono chunk");

// ---- per frame: header and bit allocation --------------------------------------------------------------------------------
// grid = (frames / 256, streams), one thread per frame.  parallel[] arrives set to kSbcRegularFlag; a frame that does not
// decode (sbc_decoder.cpp:282-295) or leaves the geometry of frame 0 clears its stream's word (= kSbcGeneralFlag).  ret: the
// reference's return value of a frame that decodes is its own affair (a frame it rejects: k_sbc_plan / k_sbc).
__global__ __launch_bounds__(256) void k_sbc_frames(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                    int n_frames, SbcFrameInfo* __restrict__ info, uint32_t* __restrict__ ret,
                                                    uint32_t* __restrict__ parallel, SbcQueues* __restrict__ queues)
{
    const int s = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 && f < 4)
        queues->extra[f] = 0;  // (k_sbc_plan: "a general stream has a chunk for the regular kernel of kind f")
    if (f >= n_frames)
        return;
    const uint8_t* base = frames + (size_t)s * stream_stride;
    const uint8_t* d = base + (size_t)f * frame_bytes;
    const uint32_t avail = (uint32_t)(n_frames - f) * (uint32_t)frame_bytes;  // bytes of the stream from this frame on
    uint32_t wb[4] = {0, 0, 0, 0}, wp[4] = {0, 0, 0, 0};
    uint32_t per_block = 0, framelen = 0xFFFF, h1 = 0, bitpool = 0, fl = 0, decoded = 0;
    // header and scale factors: twelve byte loads issued together (a byte beyond the stream's frames reads as zero -- loaded
    // from a clamped address, not branched around: every branch was a round trip to memory of its own)
    uint32_t hb[12];
#pragma unroll
    for (uint32_t i = 0; i < 12; i++) {
        const uint32_t a = d[min(i, avail - 1)];
        hb[i] = i < avail ? a : 0u;
    }
    if (frame_bytes >= 4 && hb[0] == 0x9C) {
        h1 = hb[1];
        bitpool = hb[2];
        fl = kSbcSync;
        const int mode = (h1 >> 2) & 3;
        // joint stereo and 4 subbands are not decoded (sbc_decoder.cpp:293-295); a bitpool the allocation loop can never
        // meet would hang the reference: rejected the same way
        if (mode != 3 && (h1 & 1) && bitpool <= 128) {
            fl |= kSbcOk;
            const int channels = mode ? 2 : 1, blocks = 4 * (int)(((h1 >> 4) & 3) + 1);
            uint32_t acc = 0;
#pragma unroll
            for (int c = 0; c < 2; c++)
                if (c < channels) {
                    uint8_t sc[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t a = hb[4 + ((c * 8 + j) >> 1)];
                        sc[j] = ((c * 8 + j) & 1) ? (a & 0xF) : (a >> 4);
                    }
                    int b[8];
                    bit_allocation((h1 >> 6) & 3, (h1 >> 1) & 1, (int)bitpool, sc, b);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        wb[c * 2 + (j >> 2)] |= (uint32_t)b[j] << (8 * (j & 3));
                        wp[c * 2 + (j >> 2)] |= acc << (8 * (j & 3));
                        acc += (uint32_t)b[j];
                    }
                }
            per_block = acc;
            framelen = 4 + (uint32_t)channels * 4 + ((uint32_t)blocks * acc + 7) / 8;
            decoded = (uint32_t)(blocks * 8 * channels * 2);
        }
    }
    uint4* o = reinterpret_cast<uint4*>(info + (size_t)s * n_frames + f);
    o[0] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
    o[1] = make_uint4(wp[0], wp[1], wp[2], wp[3]);
    o[2] = make_uint4(per_block | (framelen << 16), h1 | (bitpool << 8) | (fl << 16), 0u, 0u);
    const uint32_t g0 = frame_bytes >= 4 ? base[1] : 0u;
    const bool regular = (fl & kSbcOk) && ((h1 >> 4) & 3) == ((g0 >> 4) & 3) && (((h1 >> 2) & 3) != 0) == (((g0 >> 2) & 3) != 0);
    if (!regular)
        atomicAnd(&parallel[s], kSbcGeneralFlag);
    if (ret && (fl & kSbcOk))
        ret[(size_t)s * n_frames + f] = framelen | (decoded << 16);
}

namespace {

// Wave-wide inclusive scans (64 lanes) as six DPP steps: row_shr 1, 2, 4, 8 inside the rows of sixteen lanes, then row_bcast 15
// and 31 carry the rows' totals on (gfx9; a lane without a source takes the identity).  __shfl_up() would be a ds_bpermute
// round trip per step -- fifteen scans a tile: k_sbc_plan took 58 us for 256 streams.
template <int kCtrl, int kRowMask>
__device__ inline int sbc_dpp(int identity, int v)
{
    return __builtin_amdgcn_update_dpp(identity, v, kCtrl, kRowMask, 0xF, false);
}
__device__ inline int scan_max(int v, int)  // (values >= -1: -1 is the identity)
{
    v = max(v, sbc_dpp<0x111, 0xF>(-1, v));
    v = max(v, sbc_dpp<0x112, 0xF>(-1, v));
    v = max(v, sbc_dpp<0x114, 0xF>(-1, v));
    v = max(v, sbc_dpp<0x118, 0xF>(-1, v));
    v = max(v, sbc_dpp<0x142, 0xA>(-1, v));
    v = max(v, sbc_dpp<0x143, 0xC>(-1, v));
    return v;
}
__device__ inline uint32_t scan_add(uint32_t u, int)
{
    int v = (int)u;
    v += sbc_dpp<0x111, 0xF>(0, v);
    v += sbc_dpp<0x112, 0xF>(0, v);
    v += sbc_dpp<0x114, 0xF>(0, v);
    v += sbc_dpp<0x118, 0xF>(0, v);
    v += sbc_dpp<0x142, 0xA>(0, v);
    v += sbc_dpp<0x143, 0xC>(0, v);
    return (uint32_t)v;
}
// the value of the lane before (lane 0: `first`): wave_shr 1
__device__ inline int lane_before(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }

struct SbcGeom {
    int blocks, channels;
    bool four;  // four subbands: nothing is synthesised
};
__device__ inline SbcGeom geom_of_header(uint32_t h1) { return {4 * (int)(((h1 >> 4) & 3) + 1), ((h1 >> 2) & 3) ? 2 : 1, (h1 & 1) == 0}; }
// the geometry a decoder state holds, as k_sbc reads it (a state buffer that was not zero-initialised must not index out of range)
__device__ inline SbcGeom geom_of_state(const SbcState* st) { return {min((int)st->blocks, 16), min((int)st->channels, 2), st->subbands == 4}; }

}  // namespace

// grid = streams + 1, block = 256.
// The LAST workgroup sorts the streams onto the work lists, a thread per stream (a stream's place on its list from ballots and a
// sum over the waves -- an atomic per stream on the three list counters was 14 us for 1024 streams).
// Workgroup s < streams, its first wave: the plan of stream s if it is a general one.
__global__ __launch_bounds__(256) void k_sbc_plan(const SbcFrameInfo* __restrict__ info, int n_frames, int flags,
                                                  const SbcState* __restrict__ states, SbcFramePlan* __restrict__ plan,
                                                  uint32_t* __restrict__ parallel, SbcQueues* __restrict__ queues,
                                                  uint32_t* __restrict__ lists, int n_streams, int list_stride,
                                                  uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                                  SbcExtraItem* __restrict__ extra, uint8_t* __restrict__ cover)
{
    if ((int)blockIdx.x == n_streams) {
        __shared__ uint32_t wsum[3][4];
        __shared__ uint32_t base[3];
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        if (tid < 3)
            base[tid] = 0;
        __syncthreads();
        for (int s0 = 0; s0 < n_streams; s0 += 256) {
            const int s = s0 + tid;
            int cls = -1;
            uint32_t bcode = 0;  // block count of a regular stream's frames (4, 8, 12, 16), packed into its list entry
            if (s < n_streams) {
                if (parallel[s] == kSbcRegularFlag) {
                    const uint32_t h1 = info[(size_t)s * n_frames].h1;
                    cls = ((h1 >> 2) & 3) ? kSbcStereo : kSbcMono;
                    bcode = (h1 >> 4) & 3;
                } else if (geom_of_state(states + s).blocks & 3)
                    cls = 3;  // (k_sbc_finish: one wave, frame by frame)
                else
                    cls = kSbcGeneral;
            }
            uint32_t rank = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const unsigned long long m = __ballot(cls == c);
                if (cls == c)
                    rank = (uint32_t)__popcll(m & ((1ull << lane) - 1));
                if (lane == 0)
                    wsum[c][wave] = (uint32_t)__popcll(m);
            }
            __syncthreads();
            if (cls >= 0 && cls < 3) {
                uint32_t off = base[cls];
                for (int w = 0; w < wave; w++)
                    off += wsum[cls][w];
                lists[(size_t)cls * list_stride + off + rank] = (uint32_t)s | (cls == kSbcGeneral ? 0u : bcode << 30);
            }
            __syncthreads();
            if (tid < 3)
                base[tid] += wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
            __syncthreads();
        }
        if (tid < 3)
            queues->count[tid] = base[tid];
        return;
    }
    const int s = blockIdx.x, lane = threadIdx.x;
    if (lane >= 64 || parallel[s] != kSbcGeneralFlag)
        return;
    const SbcFrameInfo* inf = info + (size_t)s * n_frames;
    const SbcGeom gs = geom_of_state(states + s);
    if (gs.blocks & 3) {
        // No call leaves such a state: a frame before the first header of the call would put 1-3 rows on a timeline, and the
        // nine rows before a chunk could lie in more frames than a workgroup looks back.  (The classifier, which runs beside
        // this workgroup, reads the same state and keeps the stream off the lists.)
        if (lane == 0)
            parallel[s] = kSbcSerialFlag;
        return;
    }
    const int probe = flags & 1;
    const int F = n_frames + probe;
    SbcFramePlan* pl = plan + (size_t)s * (n_frames + 1);
    const int n_gran = (n_frames + kSbcChunk - 1) / kSbcChunk;
    uint8_t* cov = cover + (size_t)s * n_gran;
    int c_gkey = -1, c_back0 = -1, c_back1 = -1, c_lb = -1;
    int c_src[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        c_src[k] = -1;
    uint32_t c_vb0 = 0, c_vb1 = 0, c_pcm = 0, c_misc = 0;
    // (a frame's word: h1 | bitpool << 8 | flags << 16; the next tile's are on their way while this one's scans run)
    auto misc_of = [&](int v) -> uint32_t { return v < F ? reinterpret_cast<const uint32_t*>(inf + max(v - probe, 0))[9] : 0u; };
    uint32_t misc_ahead = misc_of(lane);
    for (int v0 = 0; v0 < F; v0 += 64) {
        const int v = v0 + lane;
        const bool in = v < F;
        const int f = max(v - probe, 0);
        const uint32_t misc = misc_ahead;
        misc_ahead = misc_of(v + 64);
        const bool sync = (misc >> 16) & kSbcSync, ok = (misc >> 16) & kSbcOk;
        // the frame whose header left the geometry, with that header in the key's low byte (no second look at memory)
        const int gkey = max(c_gkey, scan_max(sync ? (int)((uint32_t)v << 8 | (misc & 0xFF)) : -1, lane));
        const int gsrc = gkey >> 8;
        SbcGeom g = gs;
        if (in && gkey >= 0)
            g = geom_of_header((uint32_t)gkey & 0xFF);
        // where the run of frames that decode under one geometry, this frame its last so far, began (granules below)
        const uint32_t prev = (uint32_t)lane_before((int)misc, (int)c_misc);
        const bool runs_on = ok && ((prev >> 16) & kSbcOk) && ((misc ^ prev) & 0x30) == 0 && (((misc >> 2) & 3) != 0) == (((prev >> 2) & 3) != 0);
        const int lb = max(c_lb, scan_max(runs_on ? -1 : v, lane));
        const bool synth = in && !g.four;
        const uint32_t nb0 = synth && g.channels > 0 ? (uint32_t)g.blocks : 0u, nb1 = synth && g.channels > 1 ? (uint32_t)g.blocks : 0u;
        const uint32_t pcm = synth ? (uint32_t)(g.blocks * 8 * g.channels) : 0u;
        const uint32_t pcm_c = (probe && v == 0) ? 0u : pcm;  // (the probe's PCM is dropped)
        const uint32_t i_pcm = scan_add(pcm_c, lane), i_vb0 = scan_add(nb0, lane), i_vb1 = scan_add(nb1, lane);
        const int i_back0 = scan_max(nb0 ? v : -1, lane), i_back1 = scan_max(nb1 ? v : -1, lane);
        // exclusive: what lies before this frame
        const int e_back0 = max(lane_before(i_back0, -1), c_back0), e_back1 = max(lane_before(i_back1, -1), c_back1);
        const SbcGeom hg = geom_of_header(misc & 0xFF);
        int src[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool writes = ok && hg.blocks > 4 * (k >> 1) && hg.channels > (k & 1);
            src[k] = max(c_src[k], scan_max(writes ? v : -1, lane));
        }
        if (in) {
            uint4* o = reinterpret_cast<uint4*>(pl + v);
            o[0] = make_uint4((uint32_t)src[0], (uint32_t)src[1], (uint32_t)src[2], (uint32_t)src[3]);
            o[1] = make_uint4((uint32_t)src[4], (uint32_t)src[5], (uint32_t)src[6], (uint32_t)src[7]);
            o[2] = make_uint4(c_pcm + i_pcm - pcm_c, c_vb0 + i_vb0 - nb0, c_vb1 + i_vb1 - nb1, (uint32_t)e_back0);
            o[3] = make_uint4((uint32_t)e_back1, (uint32_t)gsrc, (uint32_t)g.blocks | ((uint32_t)g.channels << 8) | (synth ? 1u << 16 : 0u),
                              (uint32_t)lb);
            if (ret && !ok && v >= probe)
                ret[(size_t)s * n_frames + f] = 0xFFFFu | ((pcm * 2) << 16);
        }
        // ---- granules a regular kernel can take --------------------------------------------------------------------------------------
        // In units of eight frames (k_sbc_gen's items; a tile holds eight): a granule is REGULAR when its frames and the frames
        // that hold the nine blocks before it all decode under one geometry -- its last frame decodes and the run it ends began
        // early enough.  Then every sample in reach is the frame's own, the rows before the granule are the last blocks of the
        // frames before it, and only the PCM offset is not what a regular stream's would be.  Not the first granule (the state's
        // filter memory), not the last frame's (it leaves the state), not with the probe.  A stereo granule goes to
        // k_sbc_par_stereo as it is, two mono granules that make an aligned 16 frames to k_sbc_par_mono; every slot of the two
        // tables (stream x chunk of the kernel's size) and every cover byte is written, taken or not.  The last lane of a
        // granule decides (lanes 7 and 15 of a row of sixteen; what they need of each other comes over DPP row shifts).
        {
            const uint32_t pcm_at = c_pcm + i_pcm - pcm_c;  // PCM samples before this frame
            const int gi = v >> 3;                          // (probe: no granule is taken, v - f does not matter)
            const uint32_t h1 = misc & 0xFF;
            const int pre = (9 + 4 * (int)((h1 >> 4) & 3) + 3) / (4 * (int)(((h1 >> 4) & 3) + 1));  // frames that hold nine blocks
            const bool last_of_gran = (lane & 7) == 7;
            const bool reg = last_of_gran && !probe && gi >= 1 && v + 1 < n_frames && ok && lb <= v - 7 - pre;
            const bool stereo = ((h1 >> 2) & 3) != 0;
            const uint32_t mine = (reg ? 1u : 0u) | (stereo ? 2u : 0u);
            const uint32_t even = (uint32_t)sbc_dpp<0x118, 0xF>(0, (int)mine);  // lane 15 of a row: what lane 7 found
            const uint32_t pcm7 = (uint32_t)sbc_dpp<0x117, 0xF>(0, (int)pcm_at), pcm15 = (uint32_t)sbc_dpp<0x11F, 0xF>(0, (int)pcm_at);
            const bool take_stereo = mine == 3;
            const bool take_mono = kSbcMonoGranules == 1 ? mine == 1 : ((lane & 15) == 15 && mine == 1 && even == 1);
            const uint32_t entry = (uint32_t)s | (((h1 >> 4) & 3) << 30);
            if (take_stereo)
                queues->extra[kSbcStereo] = 1;  // (plain stores of the same value: a kernel whose flag stays 0 does not walk the slots)
            if (take_mono)
                queues->extra[kSbcMono] = 1;
            if (last_of_gran && gi < n_gran) {
                SbcExtraItem e;
                e.entry = take_stereo ? entry : kSbcSkipEntry;
                e.chunk = (uint32_t)gi;
                e.pcm_base = pcm7;
                e.pad = 0;
                extra[(size_t)kSbcStereo * n_streams * n_gran + (size_t)s * n_gran + gi] = e;
                if (kSbcMonoGranules == 1) {
                    e.entry = take_mono ? entry : kSbcSkipEntry;
                    extra[(size_t)kSbcMono * n_streams * n_gran + (size_t)s * n_gran + gi] = e;
                    cov[gi] = (take_mono || take_stereo) ? 1 : 0;
                }
            }
            if (kSbcMonoGranules == 2 && (lane & 15) == 15 && gi - 1 < n_gran) {
                // (lane 15 of a row: the odd granule gi and the even one before it)
                SbcExtraItem e;
                e.entry = take_mono ? entry : kSbcSkipEntry;
                e.chunk = (uint32_t)(gi >> 1);
                e.pcm_base = pcm15;
                e.pad = 0;
                extra[(size_t)kSbcMono * n_streams * n_gran + (size_t)s * ((n_gran + 1) >> 1) + (gi >> 1)] = e;
                cov[gi - 1] = (take_mono || even == 3) ? 1 : 0;
                if (gi < n_gran)
                    cov[gi] = (take_mono || take_stereo) ? 1 : 0;
            }
        }
        // carries: the last lane's inclusive values
        c_gkey = __builtin_amdgcn_readlane(gkey, 63);
        c_lb = __builtin_amdgcn_readlane(lb, 63);
        c_misc = (uint32_t)__builtin_amdgcn_readlane((int)misc, 63);
        c_back0 = max(c_back0, __builtin_amdgcn_readlane(i_back0, 63));
        c_back1 = max(c_back1, __builtin_amdgcn_readlane(i_back1, 63));
        c_pcm += (uint32_t)__builtin_amdgcn_readlane((int)i_pcm, 63);
        c_vb0 += (uint32_t)__builtin_amdgcn_readlane((int)i_vb0, 63);
        c_vb1 += (uint32_t)__builtin_amdgcn_readlane((int)i_vb1, 63);
#pragma unroll
        for (int k = 0; k < 8; k++)
            c_src[k] = __builtin_amdgcn_readlane(src[k], 63);
    }
    if (pcm_count && lane == 0)
        pcm_count[s] = c_pcm;
}

// ---- the frame-parallel kernels ----------------------------------------------------------------------------------------------
constexpr int kSbcStageDwords = 1024;        // 4 KB of frame bytes in LDS
constexpr int kSbcRowPitch = EFX_SBC_PITCH;    // dwords between two rows of matrixing outputs in LDS
constexpr uint32_t kSbcSlack = 528;          // a frame's bit fields may run past the frame size the caller states (at most 524 bytes
                                             // from its start): the reference reads on into the next frame's bytes, so does this

namespace {

// Work item `it` of this workgroup on list `cls`: the items of a list are dealt round robin (they cost the same, but for a
// stream's last chunk; a shared counter was tried first: 48 k atomics on one address serialise at 50-90 per microsecond and
// took longer than the decoding).  false when the list is through.
__device__ inline bool sbc_next_item(const SbcQueues* queues, const uint32_t* lists, int cls, int n_streams, int chunks, int it, int* s,
                                     int* chunk)
{
    __syncthreads();  // (the LDS of the previous item is free)
    const uint32_t item = blockIdx.x + (uint32_t)it * gridDim.x;
    if (item >= queues->count[cls] * (uint32_t)chunks)
        return false;
    *s = (int)lists[(size_t)cls * n_streams + item / (uint32_t)chunks];
    *chunk = (int)(item % (uint32_t)chunks);
    return true;
}

// IQUANT (sbc_decoder.cpp:263-270): sample = ((v << 1 | 1) << scale) / (2^bits - 1) - (1 << scale), the shift wrapping at 32
// bits and the division C's (truncating, of the wrapped value as a signed number)
__device__ inline int32_t sbc_iquant(uint32_t v, uint32_t bits, uint32_t scale, uint32_t magic)
{
    const uint32_t n = ((v << 1) | 1u) << scale;
    const bool neg = (int32_t)n < 0;
    const uint32_t a = neg ? 0u - n : n;
    const uint32_t t = __umulhi(magic, a);
    const uint32_t q = (t + ((a - t) >> (bits > 1 ? 1 : 0))) >> ((bits - 1) & 31);
    return (neg ? -(int32_t)q : (int32_t)q) - (int32_t)(1u << scale);
}

// acc + a * b with 24-bit operands, as ONE v_mad_i32_i24: written as __mul24() + add, the sums of a dot product are
// re-associated into multiplies and three-input adds -- 1.4 instructions per term (hipcc -S) in kernels the VALU bounds
__device__ inline uint32_t mad24(int32_t a, int32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}

// A workgroup barrier for LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits for every global
// load in flight (vmcnt(0)) -- among them the ones the frame-parallel kernel issues for its NEXT work item.
__device__ inline void sbc_lds_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

}  // namespace

// Regular streams: every frame decodes under the geometry of frame 0, so everything has a closed form -- frame f's PCM sits at
// f * its size, block b of frame f is row f * blocks + b of the channel's timeline.  One instantiation per channel count (a
// mono stream needs half the LDS).  Values multiplied with v_mad_i32_i24: table entries (18 bits), subband samples (|s| < 2^18:
// IQUANT's quotient) and matrixing outputs (an int32 >> 15) all fit 24 bits -- the low 32 bits of the products are the
// reference's, at four times the rate of a full 32-bit multiply.  (Which holds for the filter memory of a state, too, as long as
// it is one a call of this library left, or zeros.)
template <int C, int CH>
__device__ __forceinline__ void sbc_par_body(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes, int n_frames,
                                             const SbcState* __restrict__ states, SbcState* __restrict__ states_next,
                                             const SbcTables* __restrict__ tables, const SbcFrameInfo* __restrict__ info,
                                             int16_t* __restrict__ pcm, size_t pcm_stride, uint32_t* __restrict__ pcm_count, int flags,
                                             SbcQueues* __restrict__ queues, const uint32_t* __restrict__ lists, int n_streams,
                                             const SbcExtraItem* __restrict__ extra)
{
    constexpr int PB = 8 * C;  // samples of a block
    __shared__ SbcTables tb;
    constexpr int FR = CH + 3;  // + the frames that hold the nine blocks before them (blocks >= 4)
    __shared__ int32_t sb[FR * 16 * PB];              // dequantised samples of the frames in reach: [k][blk][c][sb]
    // matrixing outputs: nine blocks of history + the chunk's.  Rows are kSbcRowPitch dwords apart: the windowing reads column o
    // of rows four apart in the lanes of a group (ds_read_b32 banks = dword address mod 32; a pitch of 16 put four lanes on
    // every bank it touched)
    __shared__ int32_t rows[C][9 + CH * 16][kSbcRowPitch];
    __shared__ uint32_t sh_in[kSbcStageDwords];
    __shared__ SbcFrameInfo sh_info[FR];
    __shared__ uint32_t sh_meta[FR][PB];              // per (frame, channel, subband): bits | prefix << 8 | scale factor << 16
    __shared__ uint2 sh_fk[FR];                       // per frame: where its sample bits start in the stream (in bits), bits per block

    const int tid = threadIdx.x;
    constexpr int cls = C == 1 ? kSbcMono : kSbcStereo;
    const uint32_t count = queues->count[cls], n_general = queues->extra[cls] ? queues->count[kSbcGeneral] : 0u;
    if (count + n_general == 0)
        return;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tb);
        for (int i = tid; i < (int)(sizeof(SbcTables) / 4); i += 256)
            dst[i] = src[i];
    }
    const uint32_t chunks = (uint32_t)(n_frames + CH - 1) / CH, n_listed = count * chunks, total = n_listed + n_general * chunks;
    const uint32_t limit = (uint32_t)n_frames * (uint32_t)frame_bytes;
    const bool probe = (flags & 1) != 0;  // decode_audio()'s frame-size probe: frame 0 is decoded once more up front

    // Work items = (stream of the list, chunk of CH frames), dealt round robin: this workgroup's are blockIdx.x + it * gridDim.x.
    // A lane keeps the list entry and the chunk of one of the next 64 (one vector load per 64 items; readlane hands them out).
    // Behind the listed streams' items: the chunks of the GENERAL streams -- those of them that k_sbc_plan found regular
    // (SbcExtraItem: the same work, but the PCM of the chunk's first frame sits where the plan says: my_pcm; ~0 = at f0 * the
    // frame's samples); the others are k_sbc_gen's and skipped here.
    uint32_t my_entry = 0xFFFFFFFFu, my_chunk = 0, my_pcm = 0xFFFFFFFFu;
    auto load_entries = [&](int it0) {
        const uint32_t item = blockIdx.x + (uint32_t)(it0 + (tid & 63)) * gridDim.x;
        my_entry = my_pcm = 0xFFFFFFFFu;
        if (item < n_listed) {
            const uint32_t si = item / chunks;
            my_chunk = item - si * chunks;
            my_entry = lists[(size_t)cls * n_streams + si];
        } else if (item < total) {
            const uint32_t x = item - n_listed, gi = x / chunks;
            const SbcExtraItem e = extra[(size_t)lists[(size_t)kSbcGeneral * n_streams + gi] * chunks + (x - gi * chunks)];
            my_entry = e.entry;
            my_chunk = e.chunk;
            my_pcm = e.pcm_base;
        }
    };
    // everything about an item that is the same for all threads
    struct Item {
        bool valid, skip, staged;
        int s, blocks, f0, f1, vb0, vb1, first_vb, fr_lo, n_fr;
        uint32_t inv_b, lo_byte, hi_byte, mis, span, n_dw, n_zero, pcm_base;
        const uint8_t* gbase;
        const SbcFrameInfo* inf;
    };
    auto setup = [&](int it) -> Item {
        Item p;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)my_entry, it & 63);
        const int chunk = __builtin_amdgcn_readlane((int)my_chunk, it & 63);
        p.pcm_base = (uint32_t)__builtin_amdgcn_readlane((int)my_pcm, it & 63);
        p.valid = e != 0xFFFFFFFFu;
        p.skip = e == kSbcSkipEntry;
        const bool work = p.valid && !p.skip;
        p.s = work ? (int)(e & 0x3FFFFFFFu) : 0;
        p.blocks = 4 * (int)(((work ? e : 0u) >> 30) + 1);  // (k_sbc_plan packs the block count of the stream's frames into the entry)
        // x / blocks = x * inv_b >> 16, exact for x < 4096 ONLY: it is applied to block numbers counted from the first frame
        // in reach (< FR * 16), never to a block of the call's timeline -- that grows with the number of frames of the call
        // (round-5 ADVICE: with the absolute number the quotient was first wrong at frame 688 of a 12-block stream)
        p.inv_b = p.blocks == 4 ? 16385u : p.blocks == 8 ? 8193u : p.blocks == 12 ? 5462u : 4097u;
        p.gbase = frames + (size_t)p.s * stream_stride;
        p.inf = info + (size_t)p.s * n_frames;
        p.f0 = chunk * CH;
        p.f1 = min(n_frames, p.f0 + CH);
        // Virtual block timeline: block vb >= 0 is block vb % blocks of frame vb / blocks; with the probe, blocks
        // -blocks .. -1 are frame 0's once more; everything before comes from the state's nine history rows.
        p.vb0 = p.f0 * p.blocks;
        p.vb1 = p.f1 * p.blocks;
        p.first_vb = max(p.vb0 - 9, probe ? -p.blocks : 0);      // first block whose samples this workgroup needs
        // ... it lies in this frame (-1: the probe's frame 0).  first_vb >= (f0 - 3) * blocks: the quotient is f0 minus
        // what the (at most nine) blocks below vb0 make in frames -- a division of a number below 16, exact
        p.fr_lo = p.first_vb < 0 ? -1 : p.f0 - (int)(((uint32_t)(p.vb0 - p.first_vb + p.blocks - 1) * p.inv_b) >> 16);
        p.n_fr = p.f1 - p.fr_lo;                                   // frames in reach (<= FR)
        // the frame bytes in reach (contiguous in the stream), whole aligned dwords, bytes beyond the stream's frames as zeros
        p.lo_byte = (uint32_t)max(p.fr_lo, 0) * (uint32_t)frame_bytes;
        p.hi_byte = min((uint32_t)p.f1 * (uint32_t)frame_bytes + kSbcSlack, limit);
        p.mis = (uint32_t)((uintptr_t)(p.gbase + p.lo_byte) & 3);
        p.span = p.hi_byte - p.lo_byte + p.mis;
        p.n_dw = (p.span + 3) / 4;
        // (the stream ends inside the reach: zeros behind its last byte, as far as a frame's bit fields can run -- the
        // sample loop then reads the stage without a look at the limit)
        p.n_zero = p.hi_byte == limit ? kSbcSlack / 4 + 2 : 1;
        p.staged = p.n_dw + p.n_zero <= (uint32_t)kSbcStageDwords;
        return p;
    };
    // The global loads of an item -- its frame bytes and the frames' SbcFrameInfo -- are ISSUED while the item before it is
    // being decoded and land in registers; they go to LDS when their item's turn comes.  (Whole aligned dwords: the first and
    // the last may hold bytes that are not the reach's -- an aligned dword with one valid byte lies in that byte's page --
    // and are masked on their way to LDS.)  Nothing between issue and use waits for memory: the barriers of the loop are
    // LDS barriers (sbc_lds_barrier), not __syncthreads().
    struct Fetch {
        uint32_t w[kSbcStageDwords / 256];
        uint4 info;
    };
    auto fetch = [&](const Item& p) -> Fetch {
        Fetch r = {};
        if (!p.valid || p.skip)
            return r;
        if (p.staged) {
            const uint32_t* a0 = reinterpret_cast<const uint32_t*>(p.gbase + p.lo_byte - p.mis);
#pragma unroll
            for (int j = 0; j < kSbcStageDwords / 256; j++)
                if ((uint32_t)(tid + 256 * j) < p.n_dw)
                    r.w[j] = a0[tid + 256 * j];
        }
        if (tid < p.n_fr * 3)
            r.info = reinterpret_cast<const uint4*>(p.inf + max(p.fr_lo + tid / 3, 0))[tid % 3];
        return r;
    };
    auto commit = [&](const Item& p, const Fetch& r) {
        if (p.staged) {
#pragma unroll
            for (int j = 0; j < kSbcStageDwords / 256; j++) {
                const uint32_t i = (uint32_t)(tid + 256 * j);
                if (i < p.n_dw + p.n_zero) {
                    uint32_t w = i < p.n_dw ? r.w[j] : 0u;
                    if (i * 4 + 4 > p.span)  // the last dword: bytes beyond the reach (or the stream's frames) are zeros
                        w &= i * 4 < p.span ? 0xFFFFFFFFu >> (8 * (i * 4 + 4 - p.span)) : 0u;
                    if (i == 0)              // the first: bytes before the reach
                        w &= 0xFFFFFFFFu << (8 * p.mis);
                    sh_in[i] = __builtin_amdgcn_perm(0u, w, 0x00010203u);  // MSB first: a bit field is a shift away
                }
            }
        }
        if (tid < p.n_fr * 3)
            reinterpret_cast<uint4*>(&sh_info[tid / 3])[tid % 3] = r.info;
    };

    load_entries(0);
    Item cur = setup(0);
    Fetch got = fetch(cur);
    for (int it = 0; cur.valid; it++) {
        if (cur.skip) {  // (k_sbc_gen's chunk)
            if (((it + 1) & 63) == 0)
                load_entries(it + 1);
            cur = setup(it + 1);
            got = fetch(cur);
            continue;
        }
        sbc_lds_barrier();  // (the LDS of the previous item is free)
        commit(cur, got);
        if (((it + 1) & 63) == 0)
            load_entries(it + 1);
        const Item nxt = setup(it + 1);
        got = fetch(nxt);
        const Item p = cur;
        cur = nxt;
        const int s = p.s, blocks = p.blocks, f0 = p.f0, f1 = p.f1, vb0 = p.vb0, vb1 = p.vb1, first_vb = p.first_vb, fr_lo = p.fr_lo,
                  n_fr = p.n_fr;
        const uint32_t inv_b = p.inv_b, lo_byte = p.lo_byte, mis = p.mis;
        const bool staged = p.staged;
        // block vb >= 0 of the call's timeline, counted from the first frame in reach: frame fr_base + rel / blocks, rel < FR * 16
        const int fr_base = max(fr_lo, 0), vb_base = fr_base * blocks, k_base = fr_base - fr_lo;
        const uint8_t* gbase = p.gbase;
        // byte `pos` of the stream (bytes at or beyond `limit` read as zero); the stage holds its dwords MSB first
        auto byte_at = [&](uint32_t pos) -> uint32_t {
            if (staged)
                return reinterpret_cast<const uint8_t*>(sh_in)[(pos - lo_byte + mis) ^ 3];
            return pos < limit ? gbase[pos] : 0u;
        };
        // `bits` (1..16) bits from bit `bitpos` on: of the stage (bitpos counts from the stage's first bit) ...
        auto field_staged = [&](uint32_t bitpos, uint32_t bits) -> uint32_t {
            const uint32_t di = bitpos >> 5;
            const uint64_t x = ((uint64_t)sh_in[di] << 32) | sh_in[di + 1];
            return (uint32_t)((x << (bitpos & 31)) >> 32) >> ((32 - bits) & 31);
        };
        // ... or of the stream (a reach too fat for the stage)
        auto field_global = [&](uint32_t bitpos, uint32_t bits) -> uint32_t {
            const uint32_t pos = bitpos >> 3;
            uint32_t w = 0;
            for (uint32_t k = 0; k < 4; k++)
                w = (w << 8) | (pos + k < limit ? gbase[pos + k] : 0u);
            return (w << (bitpos & 7)) >> ((32 - bits) & 31);
        };
        sbc_lds_barrier();
        // ---- scale factors: one thread per (frame, channel, subband) ----------------------------------------------------------
        for (int i = tid; i < n_fr * PB; i += 256) {
            const int k = i / PB, r = i - k * PB;
            const uint32_t fpos = (uint32_t)max(fr_lo + k, 0) * (uint32_t)frame_bytes;
            const uint32_t a = byte_at(fpos + 4 + (uint32_t)(r >> 1));
            const uint8_t* fi = reinterpret_cast<const uint8_t*>(&sh_info[k]);
            sh_meta[k][r] = fi[r] | ((uint32_t)fi[16 + r] << 8) | (((r & 1) ? (a & 0xF) : (a >> 4)) << 16);
            if (r == 0)
                sh_fk[k] = make_uint2((fpos + 4 + PB / 2 - (staged ? lo_byte - mis : 0u)) * 8, sh_info[k].per_block);
        }
        // ---- history rows that predate every frame in reach: the state's --------------------------------------------------------
        // row t of `rows` is virtual block vb0 - 9 + t
        const SbcState* st = states + s;
        const int hist_end = probe ? -blocks : 0;  // virtual blocks below this one are the state's history: block v -> hist[9 + v - hist_end]
        for (int i = tid; i < C * 9 * 16; i += 256) {
            const int c = i / 144, r = i - c * 144, t = r >> 4, o = r & 15, vb = vb0 - 9 + t;
            if (vb < hist_end)
                rows[c][t][o] = st->hist[c][9 + vb - hist_end][o];
        }
        sbc_lds_barrier();
#if EFX_SBC_SKIP & 1
        sbc_lds_barrier();
#else
        // ---- samples (get_samples(), sbc_decoder.cpp:297-343): a thread takes (frame, channel, subband) and four consecutive
        // blocks -- width, place and scale factor of the field, the divisor's constant are read once, the bit position advances
        // by the block's bits.  Of the first frame in reach only the quarters that hold the nine blocks before the chunk.
        {
            const int r = tid & (PB - 1);
            const int QB = blocks >> 2;  // quarters of a frame
            const uint32_t inv_q = QB == 1 ? 65537u : QB == 2 ? 32769u : QB == 3 ? 21846u : 16385u;
            const int q_lo = fr_lo >= 0 ? (first_vb - fr_lo * blocks) >> 2 : 0, nq0 = QB - q_lo;
            const int n_tasks = nq0 + (n_fr - 1) * QB;
            auto tasks = [&](auto&& field) {
                for (int t = tid / PB; t < n_tasks; t += 256 / PB) {
                    int k = 0, q = q_lo + t;
                    if (t >= nq0) {
                        const int u = t - nq0, kq = (int)(__umul24((uint32_t)u, inv_q) >> 16);
                        k = 1 + kq;
                        q = u - kq * QB;
                    }
                    const uint32_t meta = sh_meta[k][r], bits = meta & 0xFF, scale = meta >> 16;
                    const uint2 fk = sh_fk[k];
                    const uint32_t magic = tb.iq_magic[bits];
                    uint32_t bitpos = fk.x + __umul24((uint32_t)(4 * q), fk.y) + ((meta >> 8) & 0xFF);
                    int32_t* dst = &sb[(k * 16 + 4 * q) * PB + r];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int32_t sample = sbc_iquant(field(bitpos, bits), bits, scale, magic);
                        dst[i * PB] = bits ? sample : 0;
                        bitpos += fk.y;
                    }
                }
            };
            if (staged)
                tasks(field_staged);
            else
                tasks(field_global);
        }
        sbc_lds_barrier();
#endif
        // ---- matrixing (sbc_decoder.cpp:86-104): rows[c][t][o] = (sum_j syn[o][j] * sb[blk][c][j]) >> 15.  The chunk's own
        // rows: below; the nine rows before them that are not the state's: one thread per output.
        const int n_t = 9 + (vb1 - vb0);
        for (int i = tid; i < C * 144; i += 256) {
            const int c = i / 144, rr = i - c * 144, t = rr >> 4, o = rr & 15, vb = vb0 - 9 + t;
            if (vb < hist_end)
                continue;
            const int rel = vb < 0 ? 0 : vb - vb_base;
            const int fq = (int)(__umul24((uint32_t)rel, inv_b) >> 16);
            const int k = vb < 0 ? -1 - fr_lo : fq + k_base, blk = vb < 0 ? vb + blocks : rel - (int)__umul24((uint32_t)fq, (uint32_t)blocks);
            const int32_t* x = &sb[((k * 16 + blk) * C + c) * 8];
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 8; j++)
                acc = mad24(tb.syn[o * 8 + j], x[j], acc);
            rows[c][t][o] = (int32_t)acc >> 15;
        }
#if EFX_SBC_SKIP & 2
#elif EFX_SBC_M
        {
            // outputs o8 and o8 + 8 of TWO consecutive rows (blocks of one frame: a block count is even) per trip: the 16
            // coefficients and the address arithmetic serve 32 multiply-adds
            const int o8 = tid & 7;
            int32_t sa[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                sa[j] = tb.syn[o8 * 8 + j];
                sh[j] = tb.syn[(o8 + 8) * 8 + j];
            }
            const int RH = (vb1 - vb0) >> 1;
            for (int idx = tid >> 3; idx < C * RH; idx += 32) {
                const int c = (C == 2 && idx >= RH) ? 1 : 0, row = 2 * (idx - c * RH), vb = vb0 + row;
                const int rel = vb - vb_base, fq = (int)(__umul24((uint32_t)rel, inv_b) >> 16), k = fq + k_base,
                          blk = rel - (int)__umul24((uint32_t)fq, (uint32_t)blocks);
                const int4* sp = reinterpret_cast<const int4*>(&sb[((k * 16 + blk) * C + c) * 8]);
                const int4 x0 = sp[0], x1 = sp[1], y0 = sp[2 * C], y1 = sp[2 * C + 1];
                const int32_t x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                const int32_t y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
                uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    a0 = mad24(sa[j], x[j], a0);
                    a1 = mad24(sh[j], x[j], a1);
                    b0 = mad24(sa[j], y[j], b0);
                    b1 = mad24(sh[j], y[j], b1);
                }
                rows[c][9 + row][o8] = (int32_t)a0 >> 15;
                rows[c][9 + row][o8 + 8] = (int32_t)a1 >> 15;
                rows[c][10 + row][o8] = (int32_t)b0 >> 15;
                rows[c][10 + row][o8 + 8] = (int32_t)b1 >> 15;
            }
        }
#else
        {
            const int o8 = tid & 7;
            int32_t sa[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                sa[j] = tb.syn[o8 * 8 + j];
                sh[j] = tb.syn[(o8 + 8) * 8 + j];
            }
            const int RB = vb1 - vb0;
            for (int idx = tid >> 3; idx < C * RB; idx += 32) {
                const int c = (C == 2 && idx >= RB) ? 1 : 0, row = idx - c * RB, vb = vb0 + row;
                const int rel = vb - vb_base, fq = (int)(__umul24((uint32_t)rel, inv_b) >> 16), k = fq + k_base,
                          blk = rel - (int)__umul24((uint32_t)fq, (uint32_t)blocks);
                const int4* sp = reinterpret_cast<const int4*>(&sb[((k * 16 + blk) * C + c) * 8]);
                const int4 x0 = sp[0], x1 = sp[1];
                const int32_t x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                uint32_t a0 = 0, a1 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    a0 = mad24(sa[j], x[j], a0);
                    a1 = mad24(sh[j], x[j], a1);
                }
                rows[c][9 + row][o8] = (int32_t)a0 >> 15;
                rows[c][9 + row][o8 + 8] = (int32_t)a1 >> 15;
            }
        }
#endif
        sbc_lds_barrier();
        // ---- windowing (sbc_decoder.cpp:106-137): sample o of block t from rows t .. t - 9; a thread makes sample o of four
        // consecutive blocks (a frame's block count is a multiple of four) out of one pass over the thirteen rows they read
        int16_t* out = pcm + (size_t)s * pcm_stride;
        const int frame_samples = C * blocks * 8;
        const size_t pcm0 = p.pcm_base != 0xFFFFFFFFu ? (size_t)p.pcm_base : (size_t)f0 * frame_samples;  // where frame f0's PCM goes
#if !(EFX_SBC_SKIP & 4)
        {
            const int o = tid & 7;
            int32_t p[10];
#pragma unroll
            for (int j = 0; j < 10; j++)
                p[j] = tb.proto[o * 10 + j];
            const int segs_c = (vb1 - vb0) >> 2;
            for (int seg = tid >> 3; seg < C * segs_c; seg += 32) {
                const int c = (C == 2 && seg >= segs_c) ? 1 : 0, r0 = (seg - c * segs_c) * 4;
                const int32_t* R = &rows[c][r0][0];  // (row 9 + r0 is the first of the four: R is the row nine before it)
                int32_t lo[13], hi[13];
#pragma unroll
                for (int j = 0; j < 13; j++) {
                    lo[j] = R[j * kSbcRowPitch + o];
                    hi[j] = R[j * kSbcRowPitch + o + 8];
                }
                const int fl = (int)(__umul24((uint32_t)r0, inv_b) >> 16), blk = r0 - fl * blocks;
                int16_t* op = out + pcm0 + (size_t)fl * frame_samples + c * blocks * 8 + blk * 8 + o;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int j = 0; j < 10; j += 2) {
                        acc = mad24(lo[9 + i - j], p[j], acc);
                        acc = mad24(hi[9 + i - j - 1], p[j + 1], acc);
                    }
                    int32_t v = (int32_t)acc >> 15;
                    v = v < -0x7FFF ? -0x7FFF : (v > 0x7FFF ? 0x7FFF : v);
                    op[i * 8] = (int16_t)v;
                }
            }
        }
#endif  // (ablation)
        // ---- the workgroup of the last frames leaves the decoder state -------------------------------------------------------------
        // NOT in `states`: the workgroup of the stream's first frames reads the history there, in this same launch, whenever it
        // gets to it.  The new state goes to `states_next` as a whole (what this geometry does not touch copied over) and
        // k_sbc_commit, the next kernel on the stream, puts it in place.
        if (f1 == n_frames && p.pcm_base == 0xFFFFFFFFu) {  // (a general stream's last frame is k_sbc_gen's)
            SbcState* so = states_next + s;
            const int k_last = n_frames - 1 - fr_lo;
            for (int i = tid; i < 256; i += 256) {
                const int blk = i >> 4, c = (i >> 3) & 1, j = i & 7;
                // (blocks beyond this geometry keep what an earlier frame left there: the reference's array is not cleared)
                so->sb_sample[blk][c][j] = (blk < blocks && c < C) ? sb[((k_last * 16 + blk) * C + c) * 8 + j] : st->sb_sample[blk][c][j];
            }
            for (int i = tid; i < 2 * 144; i += 256) {
                const int c = i / 144, r = i - c * 144;
                so->hist[c][r >> 4][r & 15] = c < C ? rows[c < C ? c : 0][n_t - 9 + (r >> 4)][r & 15] : st->hist[c][r >> 4][r & 15];
            }
            if (tid == 0) {
                so->reserved = st->reserved;
                for (int k = 0; k < 8; k++)
                    so->pad[k] = st->pad[k];
                const uint32_t h1 = sh_info[k_last].h1;
                so->frequency = (h1 >> 6) & 3;
                so->blocks = (uint8_t)blocks;
                so->channels = (uint8_t)C;
                so->mode = (h1 >> 2) & 3;
                so->allocation = (h1 >> 1) & 1;
                so->subbands = 8;
                so->bitpool = sh_info[k_last].bitpool;
                if (pcm_count)
                    pcm_count[s] = (uint32_t)n_frames * (uint32_t)frame_samples;
            }
        }
    }
}

// grid = as many workgroups as fit the chip (efx_sbc_decode), block = 256
__global__ __launch_bounds__(256, EFX_SBC_WAVES) void k_sbc_par_mono(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes, int n_frames,
                                                      const SbcState* __restrict__ states, SbcState* __restrict__ states_next,
                                                      const SbcTables* __restrict__ tables, const SbcFrameInfo* __restrict__ info,
                                                      int16_t* __restrict__ pcm, size_t pcm_stride, uint32_t* __restrict__ pcm_count,
                                                      int flags, SbcQueues* __restrict__ queues, const uint32_t* __restrict__ lists,
                                                      int n_streams, const SbcExtraItem* __restrict__ extra, int extra_cap)
{
    sbc_par_body<1, EFX_SBC_CHUNK_MONO>(frames, stream_stride, frame_bytes, n_frames, states, states_next, tables, info, pcm, pcm_stride, pcm_count, flags,
                    queues, lists, n_streams, extra + (size_t)kSbcMono * extra_cap);
}
__global__ __launch_bounds__(256) void k_sbc_par_stereo(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                        int n_frames, const SbcState* __restrict__ states,
                                                        SbcState* __restrict__ states_next, const SbcTables* __restrict__ tables,
                                                        const SbcFrameInfo* __restrict__ info, int16_t* __restrict__ pcm,
                                                        size_t pcm_stride, uint32_t* __restrict__ pcm_count, int flags,
                                                        SbcQueues* __restrict__ queues, const uint32_t* __restrict__ lists, int n_streams, const SbcExtraItem* __restrict__ extra, int extra_cap)
{
    sbc_par_body<2, kSbcChunk>(frames, stream_stride, frame_bytes, n_frames, states, states_next, tables, info, pcm, pcm_stride, pcm_count, flags,
                    queues, lists, n_streams, extra + (size_t)kSbcStereo * extra_cap);
}

// ---- general streams -------------------------------------------------------------------------------------------------------------
// A work item is eight VIRTUAL frames v0 .. v1-1 of a stream; the plan says for each of them which geometry it is synthesised
// under, where its rows sit on each channel's timeline, where its PCM goes and which frame's bits every quarter (four blocks x
// one channel) of its subband samples comes from.  SLOTS = the frames whose samples the item needs: per channel up to three
// earlier frames that hold the nine rows before the chunk's (found over plan.back: a channel's timeline skips the frames that put
// nothing on it -- mono frames for channel 1, 4-subband headers for both), then the chunk's own.
constexpr int kSbcSlots = 6 + kSbcChunk;
constexpr int kSbcRowsMax = 9 + kSbcChunk * 16;

__global__ __launch_bounds__(256) void k_sbc_gen(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes, int n_frames,
                                                 const SbcState* __restrict__ states, SbcState* __restrict__ states_next,
                                                 const SbcTables* __restrict__ tables, const SbcFrameInfo* __restrict__ info,
                                                 const SbcFramePlan* __restrict__ plan, int16_t* __restrict__ pcm, size_t pcm_stride,
                                                 int flags, SbcQueues* __restrict__ queues, const uint32_t* __restrict__ lists,
                                                 int n_streams, const uint8_t* __restrict__ cover, int cover_stride)
{
    __shared__ SbcTables tb;
    __shared__ int32_t sb[kSbcSlots][16][2][8];       // the sb_sample array as it stood after each slot's get_samples()
    __shared__ int32_t rows[2][kSbcRowsMax][16];
    __shared__ uint16_t rowmap[2][kSbcRowsMax];       // row t of channel c <- slot << 4 | block; 0xFFFF: not this workgroup's to make
    __shared__ SbcFramePlan sh_plan[kSbcSlots];
    __shared__ SbcFrameInfo sh_info[kSbcSlots];
    __shared__ uint8_t sh_scale[kSbcSlots][16];
    __shared__ int sh_slot[kSbcSlots];

    const int tid = threadIdx.x;
    if (queues->count[kSbcGeneral] == 0)
        return;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tb);
        for (int i = tid; i < (int)(sizeof(SbcTables) / 4); i += 256)
            dst[i] = src[i];
    }
    const int probe = flags & 1;
    const int F = n_frames + probe;
    const int chunks = (n_frames + 1 + kSbcChunk - 1) / kSbcChunk;  // (of virtual frames; the last may be empty)
    const uint32_t limit = (uint32_t)n_frames * (uint32_t)frame_bytes;
    int s, chunk;
    for (int it = 0; sbc_next_item(queues, lists, kSbcGeneral, n_streams, chunks, it, &s, &chunk); it++) {
        const int v0 = chunk * kSbcChunk, v1 = min(F, v0 + kSbcChunk);
        if (v0 >= F || (chunk < cover_stride && cover[(size_t)s * cover_stride + chunk]))
            continue;  // (nothing there, or a regular kernel's: k_sbc_plan)
        const uint8_t* gbase = frames + (size_t)s * stream_stride;
        const SbcFrameInfo* inf = info + (size_t)s * n_frames;
        const SbcFramePlan* pl = plan + (size_t)s * (n_frames + 1);
        const SbcState* st = states + s;
        // ---- slots -----------------------------------------------------------------------------------------------------------
        if (tid < 2) {
            int need = 9, g = pl[v0].back[tid];
            for (int hop = 0; hop < 3; hop++) {
                const bool take = need > 0 && g >= 0;
                sh_slot[tid * 3 + hop] = take ? g : -1;
                if (take) {
                    need -= (int)(pl[g].geom & 0xFF);  // (a frame on the chain has rows on this timeline)
                    g = pl[g].back[tid];
                }
            }
        } else if (tid < 2 + kSbcChunk)
            sh_slot[6 + tid - 2] = v0 + tid - 2 < v1 ? v0 + tid - 2 : -1;
        for (int i = tid; i < 2 * kSbcRowsMax; i += 256)
            (&rowmap[0][0])[i] = 0xFFFF;
        __syncthreads();
        if (tid < kSbcSlots * 7) {
            const int j = tid / 7, part = tid - j * 7, u = sh_slot[j];
            if (u >= 0) {
                if (part < 4)
                    reinterpret_cast<uint4*>(&sh_plan[j])[part] = reinterpret_cast<const uint4*>(pl + u)[part];
                else
                    reinterpret_cast<uint4*>(&sh_info[j])[part - 4] = reinterpret_cast<const uint4*>(inf + max(u - probe, 0))[part - 4];
            }
        } else if (tid < kSbcSlots * 7 + kSbcSlots * 2) {
            // scale factors, eight bytes per slot
            const int i = tid - kSbcSlots * 7, j = i >> 1, half = i & 1, u = sh_slot[j];
            if (u >= 0) {
                const uint32_t pos = (uint32_t)max(u - probe, 0) * (uint32_t)frame_bytes + 4 + 4 * (uint32_t)half;
                for (int b = 0; b < 4; b++) {
                    const uint32_t a = pos + b < limit ? gbase[pos + b] : 0u;
                    sh_scale[j][half * 8 + 2 * b] = (uint8_t)(a >> 4);
                    sh_scale[j][half * 8 + 2 * b + 1] = (uint8_t)(a & 0xF);
                }
            }
        }
        __syncthreads();
        // timeline index of row 0 of channel c, and how many rows the channel has here (nine before the chunk's first + the chunk's)
        const SbcFramePlan& pf = sh_plan[6];
        const SbcFramePlan& plast = sh_plan[6 + v1 - v0 - 1];
        int base[2], n_t[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            base[c] = (int)pf.vb[c] - 9;
            const bool has = ((plast.geom >> 16) & 1) && (int)((plast.geom >> 8) & 0xFF) > c;
            n_t[c] = (int)plast.vb[c] + (has ? (int)(plast.geom & 0xFF) : 0) - base[c];
        }
        const bool last_item = v1 == F;
        // ---- row map; history rows; samples -------------------------------------------------------------------------------------------
        for (int i = tid; i < kSbcSlots * 32; i += 256) {
            const int j = i >> 5, c = (i >> 4) & 1, blk = i & 15, u = sh_slot[j];
            if (u < 0 || (j < 6 && j / 3 != c))
                continue;  // (an earlier frame serves the channel whose chain it is on)
            const SbcFramePlan& p = sh_plan[j];
            if (!((p.geom >> 16) & 1) || (int)((p.geom >> 8) & 0xFF) <= c || blk >= (int)(p.geom & 0xFF))
                continue;
            const int t = (int)p.vb[c] + blk - base[c];
            if (t >= 0 && t < n_t[c])
                rowmap[c][t] = (uint16_t)(j << 4 | blk);
        }
        for (int i = tid; i < 2 * 9 * 16; i += 256) {
            const int c = i / 144, r = i - c * 144, t = r >> 4, o = r & 15, T = base[c] + t;
            if (T < 0)
                rows[c][t][o] = st->hist[c][9 + T][o];  // (rows before the call's first: the state's filter memory)
        }
        for (int j = 0; j < kSbcSlots; j++) {
            const int u = sh_slot[j];
            if (u < 0)
                continue;
            const int blk = tid >> 4, c = (tid >> 3) & 1, sbi = tid & 7;
            const SbcFramePlan& p = sh_plan[j];
            const bool whole = last_item && j == 6 + v1 - v0 - 1;  // (the last frame's samples become the state's: all of them)
            if (!whole && (blk >= (int)(p.geom & 0xFF) || c >= (int)((p.geom >> 8) & 0xFF) || (j < 6 && j / 3 != c)))
                continue;
            const int h = p.src[(blk >> 2) * 2 + c];
            int32_t sample;
            if (h < 0)
                sample = st->sb_sample[blk][c][sbi];
            else {
                // the frame that wrote this quarter last: this one (its info is in LDS) or an earlier one
                const int fh = max(h - probe, 0);
                uint32_t bits, prefix, per_block, h1, scale;
                if (h == u) {
                    const uint8_t* fi = reinterpret_cast<const uint8_t*>(&sh_info[j]);
                    bits = fi[c * 8 + sbi];
                    prefix = fi[16 + c * 8 + sbi];
                    per_block = sh_info[j].per_block;
                    h1 = sh_info[j].h1;
                    scale = sh_scale[j][c * 8 + sbi];
                } else {
                    const SbcFrameInfo* fi = inf + fh;
                    bits = fi->bits[c][sbi];
                    prefix = fi->prefix[c][sbi];
                    per_block = fi->per_block;
                    h1 = fi->h1;
                    const uint32_t pos = (uint32_t)fh * (uint32_t)frame_bytes + 4 + (uint32_t)((c * 8 + sbi) >> 1);
                    const uint32_t a = pos < limit ? gbase[pos] : 0u;
                    scale = (sbi & 1) ? (a & 0xF) : (a >> 4);
                }
                const uint32_t data_off = 4 + (((h1 >> 2) & 3) ? 8u : 4u);
                const uint32_t bitpos = ((uint32_t)fh * (uint32_t)frame_bytes + data_off) * 8 + (uint32_t)blk * per_block + prefix;
                const uint32_t pos = bitpos >> 3;
                uint32_t w = 0;
                for (uint32_t k = 0; k < 4; k++)
                    w = (w << 8) | (pos + k < limit ? gbase[pos + k] : 0u);
                const uint32_t v = __builtin_amdgcn_ubfe(w, 32 - (bitpos & 7) - bits, bits);
                sample = bits ? sbc_iquant(v, bits, scale, tb.iq_magic[bits]) : 0;
            }
            sb[j][blk][c][sbi] = sample;
        }
        __syncthreads();
        // ---- matrixing -------------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 2; c++)
            for (int i = tid; i < n_t[c] * 16; i += 256) {
                const int t = i >> 4, o = i & 15;
                const uint32_t m = rowmap[c][t];
                if (m == 0xFFFF)
                    continue;
                const int32_t* x = &sb[m >> 4][m & 15][c][0];
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc = mad24(tb.syn[o * 8 + j], x[j], acc);
                rows[c][t][o] = (int32_t)acc >> 15;
            }
        __syncthreads();
        // ---- windowing: items (channel, frame of the chunk, block, sample) -----------------------------------------------------------------
        int16_t* out = pcm + (size_t)s * pcm_stride;
        for (int it = 0; it < 2 * kSbcChunk * 16 * 8 / 256; it++) {
            const int i = it * 256 + tid;
            const int c = i >> 10, fr = (i >> 7) & 7, blk = (i >> 3) & 15, o = i & 7;
            if (fr >= v1 - v0)
                continue;
            const SbcFramePlan& p = sh_plan[6 + fr];
            const int blocks = (int)(p.geom & 0xFF), channels = (int)((p.geom >> 8) & 0xFF);
            if (!((p.geom >> 16) & 1) || c >= channels || blk >= blocks || (probe && v0 + fr == 0))
                continue;
            const int t = (int)p.vb[c] + blk - base[c];
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 10; j += 2) {
                acc = mad24(rows[c][t - j][o], tb.proto[o * 10 + j], acc);
                acc = mad24(rows[c][t - j - 1][o + 8], tb.proto[o * 10 + j + 1], acc);
            }
            int32_t v = (int32_t)acc >> 15;
            v = v < -0x7FFF ? -0x7FFF : (v > 0x7FFF ? 0x7FFF : v);
            out[(size_t)p.pcm_off + c * blocks * 8 + blk * 8 + o] = (int16_t)v;
        }
        // ---- the item of the last frame leaves the decoder state (in states_next: k_sbc_commit) ------------------------------------------
        if (last_item) {
            SbcState* so = states_next + s;
            const int jl = 6 + v1 - v0 - 1;
            {
                const int blk = tid >> 4, c = (tid >> 3) & 1, j = tid & 7;
                so->sb_sample[blk][c][j] = sb[jl][blk][c][j];
            }
            for (int i = tid; i < 2 * 144; i += 256) {
                const int c = i / 144, r = i - c * 144;
                so->hist[c][r >> 4][r & 15] = rows[c][n_t[c] - 9 + (r >> 4)][r & 15];
            }
            if (tid == 0) {
                so->reserved = st->reserved;
                for (int k = 0; k < 8; k++)
                    so->pad[k] = st->pad[k];
                const int g = plast.gsrc;
                if (g >= 0) {
                    const SbcFrameInfo* fi = inf + max(g - probe, 0);
                    const uint32_t h1 = fi->h1;
                    so->frequency = (h1 >> 6) & 3;
                    so->blocks = (uint8_t)(4 * (((h1 >> 4) & 3) + 1));
                    so->mode = (h1 >> 2) & 3;
                    so->channels = ((h1 >> 2) & 3) ? 2 : 1;
                    so->allocation = (h1 >> 1) & 1;
                    so->subbands = (h1 & 1) ? 8 : 4;
                    so->bitpool = fi->bitpool;
                } else {
                    so->frequency = st->frequency;
                    so->blocks = (uint8_t)min((int)st->blocks, 16);
                    so->channels = (uint8_t)min((int)st->channels, 2);
                    so->mode = st->mode;
                    so->allocation = st->allocation;
                    so->subbands = st->subbands == 4 ? 4 : 8;
                    so->bitpool = st->bitpool;
                }
            }
        }
    }
}

// grid = streams, block = 64: what ends an efx_sbc_decode call.  The state a frame-parallel kernel left in `next` takes its
// place (the last chunk of a stream must not overwrite the state its first chunk may still be reading); a stream no list took
// (k_sbc_plan) is decoded here, one wave walking its frames; and the stream's word in parallel[] goes back to
// kSbcRegularFlag, where the next call's k_sbc_frames expects it.
__global__ __launch_bounds__(64) void k_sbc_finish(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                   int n_frames, SbcState* __restrict__ states, const SbcState* __restrict__ next,
                                                   const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                                   size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                                   int flags, uint32_t* __restrict__ parallel)
{
    const int s = blockIdx.x;
    const uint32_t how = parallel[s];
    if (how == kSbcSerialFlag)
        sbc_serial_body(frames, stream_stride, frame_bytes, n_frames, states, tables, pcm, pcm_stride, ret, pcm_count, flags);
    else {
        static_assert(sizeof(SbcState) % 4 == 0, "SbcState is copied by dwords");
        const uint32_t* src = reinterpret_cast<const uint32_t*>(next + s);
        uint32_t* dst = reinterpret_cast<uint32_t*>(states + s);
        for (int i = threadIdx.x; i < (int)(sizeof(SbcState) / 4); i += 64)
            dst[i] = src[i];
    }
    if (threadIdx.x == 0)
        parallel[s] = kSbcRegularFlag;
}

}  // namespace efx
