// k_sbc.hip -- SBC audio frames -> PCM for a batch of streams (gfx950).
//
// Restates sbc_decoder() and everything below it (reference src/sbc_decoder.cpp:74-373:
// get_samples 276-344, bit_allocation 142-240, IQUANT 263-270, synthesize8 74-139) the way
// decode_audio() drives it (src/video.cpp:962-989): a stream is a run of equally sized frames
// decoded in order, the synthesis filter memory carrying over from frame to frame.
//
// Two paths.  The synthesis filter is an FIR: the matrixing outputs of a block depend on that block's samples only, a PCM
// sample on the last ten blocks' outputs -- so a stream whose frames all decode and share one geometry (k_sbc_check) is
// decoded FRAME-PARALLEL by k_sbc_par: a workgroup takes eight consecutive frames, dequantises them and the nine blocks
// before them, matrixes, windows.  What chains from frame to frame in the reference -- a rejected frame re-synthesising
// the PREVIOUS frame's samples under the geometry its header left behind -- stays with k_sbc, one wave per stream walking
// its frames; INSIDE a frame nothing is serial there either:
//   * the reference's lazy bit reader is replaced by direct addressing -- every block of a frame
//     has the same layout, so sample (blk, ch, sb) starts at bit blk * bits_per_block +
//     prefix[ch][sb] -- one lane per sample extracts, dequantises (32-bit wrapping shift, C
//     division) and stores it;
//   * the reference's sliding buffer v[170] + 16 offsets (a 10-deep history per matrix output) is
//     kept as rows: row t holds the 16 matrixing outputs of block t.  The matrixing of ALL blocks
//     of the frame is computed at once (16 x 8 MACs per block, one lane per output), then the
//     windowing of all blocks (10 MACs per PCM sample, one lane per sample) reads rows t .. t-9,
//     the first nine of which are the previous frame's;
//   * accumulation is 32-bit wrapping, >> 15 arithmetic, clip to +-0x7FFF exactly as the reference.
// Frames the reference rejects behave as it does: a bad sync byte re-synthesises the state's
// previous subband samples, joint stereo switches the geometry and synthesises stale samples, a
// 4-subband header yields nothing (and keeps yielding nothing until a good frame).
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

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
    int bitcount = 0, slicecount = 0, bitslice = max_bitneed + 1;
    do {
        bitslice--;
        bitcount += slicecount;
        slicecount = 0;
        for (int sb = 0; sb < 8; sb++) {
            if (bitneed[sb] > bitslice + 1 && bitneed[sb] < bitslice + 16)
                slicecount++;
            else if (bitneed[sb] == bitslice + 1)
                slicecount += 2;
        }
    } while (bitcount + slicecount < bitpool);
    if (bitcount + slicecount == bitpool) {
        bitcount += slicecount;
        bitslice--;
    }
    for (int sb = 0; sb < 8; sb++) {
        int b = 0;
        if (bitneed[sb] >= bitslice + 2)
            b = min(bitneed[sb] - bitslice, 16);
        bits[sb] = b;
    }
    for (int sb = 0; bitcount < bitpool && sb < 8; sb++) {
        if (bits[sb] >= 2 && bits[sb] < 16) {
            bits[sb]++;
            bitcount++;
        } else if (bitneed[sb] == bitslice + 1 && bitpool > bitcount + 1) {
            bits[sb] = 2;
            bitcount += 2;
        }
    }
    for (int sb = 0; bitcount < bitpool && sb < 8; sb++)
        if (bits[sb] < 16) {
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
__global__ __launch_bounds__(64) void k_sbc(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                            int n_frames, SbcState* __restrict__ states,
                                            const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                            size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                            int flags, const uint32_t* __restrict__ parallel)
{
    if (parallel && parallel[blockIdx.x])
        return;  // (k_sbc_par decodes this stream)
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

// ---- the frame-parallel path ---------------------------------------------------------------------------------------------
constexpr int kSbcChunk = 8;                 // frames per workgroup
constexpr int kSbcFrames = kSbcChunk + 3;    // + the frames that hold the nine blocks before them (blocks >= 4)

// one thread per frame: parallel[s] stays 1 when every frame of stream s decodes (sbc_decoder.cpp:282-295) with the
// geometry of frame 0.  parallel[] arrives set to all ones.
__global__ __launch_bounds__(256) void k_sbc_check(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                   int n_frames, uint32_t* __restrict__ parallel)
{
    const int s = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames)
        return;
    const uint8_t* base = frames + (size_t)s * stream_stride;
    const uint8_t* d = base + (size_t)f * frame_bytes;
    bool ok = frame_bytes >= 4 && d[0] == 0x9C;
    if (ok) {
        const uint32_t h1 = d[1], g0 = base[1];
        const int mode = (h1 >> 2) & 3;
        ok = mode != 3 && (h1 & 1) && d[2] <= 128;
        // same blocks, same channel count as frame 0 (the PCM of a frame then sits at f * its size)
        ok = ok && ((h1 >> 4) & 3) == ((g0 >> 4) & 3) && (mode != 0) == (((g0 >> 2) & 3) != 0);
    }
    if (!ok)
        atomicAnd(&parallel[s], 0u);
}

// grid = (chunks of kSbcChunk frames, streams), block = 256.  Same arguments and results as k_sbc.  One instantiation per
// channel count (a mono stream needs half the LDS: twice the workgroups per CU); a workgroup whose stream is of the other
// kind leaves at once.  The frame bytes in reach are copied into LDS with coalesced dword loads first: headers, scale
// factors and the samples' bit fields are then LDS reads (the byte loads from global memory were 45 % of a workgroup's life).
constexpr int kSbcStageDwords = 1536;  // 6 KB: eleven frames of up to 558 bytes (16 blocks x 2 channels x 8 subbands x 16 bits + 12 is 524)
template <int C>
__device__ __forceinline__ void sbc_par_body(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                             int n_frames, const SbcState* __restrict__ states, SbcState* __restrict__ states_next,
                                             const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                             size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                             int flags, const uint32_t* __restrict__ parallel)
{
    const int s = blockIdx.y, tid = threadIdx.x;
    if (!parallel[s] || n_frames <= 0)
        return;
    const uint8_t* gbase = frames + (size_t)s * stream_stride;
    const uint32_t g0 = gbase[1];
    const int blocks = 4 * (int)(((g0 >> 4) & 3) + 1), channels = ((g0 >> 2) & 3) ? 2 : 1, per_blk = channels * 8;
    if (channels != C)
        return;
    __shared__ SbcTables tb;
    __shared__ int32_t sb[kSbcFrames][16][C][8];            // dequantised samples of the frames in reach
    __shared__ int32_t rows[C][9 + kSbcChunk * 16][16];    // matrixing outputs: nine blocks of history + the chunk's
    __shared__ uint8_t sh_scale[kSbcFrames][C][8], sh_bits[kSbcFrames][C][8];
    __shared__ uint16_t sh_prefix[kSbcFrames][C * 8], sh_per_block[kSbcFrames];
    __shared__ uint32_t sh_in[kSbcStageDwords];

    const uint32_t limit = (uint32_t)n_frames * (uint32_t)frame_bytes;
    const bool probe = (flags & 1) != 0;  // decode_audio()'s frame-size probe: frame 0 is decoded once more up front
    const int f0 = blockIdx.x * kSbcChunk, f1 = min(n_frames, f0 + kSbcChunk);
    // Virtual block timeline: block vb >= 0 is block vb % blocks of frame vb / blocks; with the probe, blocks
    // -blocks .. -1 are frame 0's once more; everything before comes from the state's nine history rows.
    const int vb0 = f0 * blocks, vb1 = f1 * blocks;
    const int first_vb = max(vb0 - 9, probe ? -blocks : 0);      // first block whose samples this workgroup needs
    const int fr_lo = first_vb < 0 ? -1 : first_vb / blocks;     // ... it lies in this frame (-1: the probe's copy of frame 0)
    const int n_fr = f1 - fr_lo;                                 // frames in reach (<= kSbcFrames)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tb);
        for (int i = tid; i < (int)(sizeof(SbcTables) / 4); i += 256)
            dst[i] = src[i];
    }
    // the bytes of frames max(fr_lo, 0) .. f1 - 1 (contiguous in the stream): whole aligned dwords, bytes beyond the stream's
    // frames as zeros (be_bits); a chunk too fat for the stage is read from global memory as before
    const uint32_t lo_byte = (uint32_t)max(fr_lo, 0) * (uint32_t)frame_bytes, hi_byte = min((uint32_t)f1 * (uint32_t)frame_bytes, limit);
    const uint32_t mis = (uint32_t)((uintptr_t)(gbase + lo_byte) & 3), n_dw = (hi_byte - lo_byte + mis + 3) / 4;
    const bool staged = n_dw <= (uint32_t)kSbcStageDwords;
    if (staged) {
        const uint8_t* a0 = gbase + lo_byte - mis;  // (>= frames: the buffer starts dword-aligned)
        for (uint32_t i = tid; i < n_dw; i += 256) {
            uint32_t w;
            if (i * 4 + 4 <= hi_byte - lo_byte + mis)
                w = *reinterpret_cast<const uint32_t*>(a0 + i * 4);
            else {  // the last dword: never past the stream's last frame byte
                w = 0;
                for (uint32_t b = 0; b < 4; b++)
                    if (i * 4 + b < hi_byte - lo_byte + mis)
                        w |= (uint32_t)a0[i * 4 + b] << (8 * b);
            }
            sh_in[i] = w;
        }
    }
    // where frame f (max(fr_lo, 0) <= f < f1) starts: in the stage when the chunk fits it (flat loads serve both)
    auto frame_at = [&](int f) -> const uint8_t* {
        return staged ? reinterpret_cast<const uint8_t*>(sh_in) + ((uint32_t)f * (uint32_t)frame_bytes - lo_byte + mis)
                      : gbase + (size_t)f * frame_bytes;
    };
    __syncthreads();
    // ---- headers, scale factors, bit allocation: one thread per (frame, channel) ---------------------------------------
    if (tid < n_fr * 2) {
        const int k = tid >> 1, c = tid & (C - 1), f = max(fr_lo + k, 0);
        const uint8_t* d = frame_at(f);
        const uint32_t avail = limit - (uint32_t)f * (uint32_t)frame_bytes;
        if ((tid & 1) < C) {
            uint8_t sc[8];
            for (int j = 0; j < 8; j++) {
                const uint32_t i = 4 + (uint32_t)((c * 8 + j) >> 1);
                const uint32_t a = i < avail ? d[i] : 0u;
                sc[j] = ((c * 8 + j) & 1) ? (a & 0xF) : (a >> 4);
                sh_scale[k][c][j] = sc[j];
            }
            int b[8];
            bit_allocation((d[1] >> 6) & 3, (d[1] >> 1) & 1, d[2], sc, b);
            for (int j = 0; j < 8; j++)
                sh_bits[k][c][j] = (uint8_t)b[j];
        }
    }
    __syncthreads();
    if (tid < n_fr) {
        int acc = 0;
        for (int c = 0; c < channels; c++)
            for (int j = 0; j < 8; j++) {
                sh_prefix[tid][c * 8 + j] = (uint16_t)acc;
                acc += sh_bits[tid][c][j];
            }
        sh_per_block[tid] = (uint16_t)acc;
    }
    __syncthreads();
    // ---- samples: one thread per (frame, blk, ch, sb) (IQUANT, sbc_decoder.cpp:263-270) --------------------------------
    const uint32_t data_off = 4 + (uint32_t)(per_blk >> 1);
    for (int i = tid; i < n_fr * blocks * per_blk; i += 256) {
        const int k = i / (blocks * per_blk), r0 = i - k * blocks * per_blk, blk = r0 / per_blk, r = r0 - blk * per_blk;
        const int f = max(fr_lo + k, 0);
        const uint8_t* d = frame_at(f);
        const uint32_t avail = limit - (uint32_t)f * (uint32_t)frame_bytes;
        const int bits = sh_bits[k][r >> 3][r & 7];
        int32_t sample = 0;
        if (bits) {
            const uint32_t bitpos = data_off * 8 + (uint32_t)blk * sh_per_block[k] + sh_prefix[k][r];
            const int scale = sh_scale[k][r >> 3][r & 7];
            int32_t q = (int32_t)be_bits(d, avail, bitpos, bits);
            q = (q << 1) | 1;
            q = (int32_t)((uint32_t)q << scale) / ((1 << bits) - 1);
            sample = q - (1 << scale);
        }
        sb[k][blk][r >> 3][r & 7] = sample;
    }
    // ---- history rows that predate every frame in reach: the state's ------------------------------------------------------
    // row t of `rows` is virtual block vb0 - 9 + t
    const SbcState* st = states + s;
    const int hist_end = probe ? -blocks : 0;  // virtual blocks below this one are the state's history: block v -> hist[9 + v - hist_end]
    for (int i = tid; i < channels * 9 * 16; i += 256) {
        const int c = i / 144, r = i - c * 144, t = r >> 4, o = r & 15, vb = vb0 - 9 + t;
        if (vb < hist_end)
            rows[c][t][o] = st->hist[c][9 + vb - hist_end][o];
    }
    __syncthreads();
    // ---- matrixing (sbc_decoder.cpp:86-104): rows[c][t][o] = (sum_j syn[o][j] * sb[blk][c][j]) >> 15 ----------------------
    const int n_t = 9 + (vb1 - vb0);
    for (int i = tid; i < channels * n_t * 16; i += 256) {
        const int c = i / (n_t * 16), r = i - c * n_t * 16, t = r >> 4, o = r & 15, vb = vb0 - 9 + t;
        if (vb < hist_end)
            continue;
        const int f = vb < 0 ? -1 : vb / blocks, blk = vb < 0 ? vb + blocks : vb - f * blocks, k = f - fr_lo;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
            acc += (uint32_t)tb.syn[o * 8 + j] * (uint32_t)sb[k][blk][c][j];
        rows[c][t][o] = (int32_t)acc >> 15;
    }
    __syncthreads();
    // ---- windowing (sbc_decoder.cpp:106-137): sample o of block t from rows t .. t - 9 ---------------------------------------
    int16_t* out = pcm + (size_t)s * pcm_stride;
    const int frame_samples = channels * blocks * 8;
    for (int i = tid; i < (f1 - f0) * frame_samples; i += 256) {
        const int fl = i / frame_samples, r0 = i - fl * frame_samples, c = r0 / (blocks * 8), r = r0 - c * blocks * 8, blk = r >> 3,
                  o = r & 7, t = 9 + fl * blocks + blk;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 10; j += 2) {
            acc += (uint32_t)rows[c][t - j][o] * (uint32_t)tb.proto[o * 10 + j];
            acc += (uint32_t)rows[c][t - j - 1][o + 8] * (uint32_t)tb.proto[o * 10 + j + 1];
        }
        int32_t v = (int32_t)acc >> 15;
        v = v < -0x7FFF ? -0x7FFF : (v > 0x7FFF ? 0x7FFF : v);
        out[(size_t)(f0 + fl) * frame_samples + r0] = (int16_t)v;
    }
    if (ret)
        for (int fl = tid; fl < f1 - f0; fl += 256) {
            const int k = f0 + fl - fr_lo;
            const uint32_t framelen = data_off + ((uint32_t)blocks * sh_per_block[k] + 7) / 8;
            ret[(size_t)s * n_frames + f0 + fl] = (framelen & 0xFFFF) | ((uint32_t)(frame_samples * 2) << 16);
        }
    // ---- the workgroup of the last frames leaves the decoder state -------------------------------------------------------------
    // NOT in `states`: the workgroup of the stream's first frames reads the history there, in this same launch, whenever it
    // gets to it.  The new state goes to `states_next` as a whole (what this geometry does not touch copied over) and
    // k_sbc_commit, the next kernel on the stream, puts it in place.
    if (f1 == n_frames) {
        SbcState* so = states_next + s;
        const int k_last = n_frames - 1 - fr_lo;
        __syncthreads();
        for (int i = tid; i < 256; i += 256) {
            const int blk = i >> 4, c = (i >> 3) & 1, j = i & 7;
            // (blocks beyond this geometry keep what an earlier frame left there: the reference's array is not cleared)
            so->sb_sample[blk][c][j] = (blk < blocks && c < channels) ? sb[k_last][blk][c][j] : st->sb_sample[blk][c][j];
        }
        for (int i = tid; i < 2 * 144; i += 256) {
            const int c = i / 144, r = i - c * 144;
            so->hist[c][r >> 4][r & 15] = c < channels ? rows[c][n_t - 9 + (r >> 4)][r & 15] : st->hist[c][r >> 4][r & 15];
        }
        if (tid == 0) {
            so->reserved = st->reserved;
            for (int k = 0; k < 8; k++)
                so->pad[k] = st->pad[k];
            const uint8_t* d = frame_at(n_frames - 1);
            so->frequency = (d[1] >> 6) & 3;
            so->blocks = (uint8_t)blocks;
            so->channels = (uint8_t)channels;
            so->mode = (d[1] >> 2) & 3;
            so->allocation = (d[1] >> 1) & 1;
            so->subbands = 8;
            so->bitpool = d[2];
            if (pcm_count)
                pcm_count[s] = (uint32_t)n_frames * (uint32_t)frame_samples;
        }
    }
}

__global__ __launch_bounds__(256) void k_sbc_par_mono(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                      int n_frames, const SbcState* __restrict__ states,
                                                      SbcState* __restrict__ states_next,
                                                      const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                                      size_t pcm_stride, uint32_t* __restrict__ ret, uint32_t* __restrict__ pcm_count,
                                                      int flags, const uint32_t* __restrict__ parallel)
{
    sbc_par_body<1>(frames, stream_stride, frame_bytes, n_frames, states, states_next, tables, pcm, pcm_stride, ret, pcm_count, flags, parallel);
}
__global__ __launch_bounds__(256) void k_sbc_par_stereo(const uint8_t* __restrict__ frames, size_t stream_stride, int frame_bytes,
                                                        int n_frames, const SbcState* __restrict__ states,
                                                        SbcState* __restrict__ states_next,
                                                        const SbcTables* __restrict__ tables, int16_t* __restrict__ pcm,
                                                        size_t pcm_stride, uint32_t* __restrict__ ret,
                                                        uint32_t* __restrict__ pcm_count, int flags,
                                                        const uint32_t* __restrict__ parallel)
{
    sbc_par_body<2>(frames, stream_stride, frame_bytes, n_frames, states, states_next, tables, pcm, pcm_stride, ret, pcm_count, flags, parallel);
}

// the states k_sbc_par left in `next` take their place (grid = streams; only streams that went the frame-parallel way)
__global__ __launch_bounds__(256) void k_sbc_commit(SbcState* __restrict__ states, const SbcState* __restrict__ next,
                                                    const uint32_t* __restrict__ parallel)
{
    const int s = blockIdx.x;
    if (!parallel[s])
        return;
    static_assert(sizeof(SbcState) % 4 == 0, "SbcState is copied by dwords");
    const uint32_t* src = reinterpret_cast<const uint32_t*>(next + s);
    uint32_t* dst = reinterpret_cast<uint32_t*>(states + s);
    for (int i = threadIdx.x; i < (int)(sizeof(SbcState) / 4); i += 256)
        dst[i] = src[i];
}

}  // namespace efx
