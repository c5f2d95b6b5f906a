// k_sbc.hip -- SBC audio frames -> PCM for a batch of streams (gfx950).
//
// Restates sbc_decoder() and everything below it (reference src/sbc_decoder.cpp:74-373:
// get_samples 276-344, bit_allocation 142-240, IQUANT 263-270, synthesize8 74-139) the way
// decode_audio() drives it (src/video.cpp:962-989): a stream is a run of equally sized frames
// decoded in order, the synthesis filter memory carrying over from frame to frame.
//
// One wave per stream walks its frames; INSIDE a frame nothing is serial:
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

}  // namespace efx
