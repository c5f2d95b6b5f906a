// k_video.hip -- composite-video field synthesis and PDM delta-sigma modulation (gfx950).
//
// k_composite restates video_isr() and everything it calls (reference src/video.cpp:1122-1198:
// sync 889-893, burst 806-837, burst_pal 636-644, blanking 904-914, pal_sync 918-934 and the
// pixel blitter blit 690-804) as a pure function of (front frame, standard, _frame_counter):
// the ISR writes every sample of every line either directly or through the black fill the
// preceding blanking lines left in the two ping-pong DMA buffers, so a whole field is
// data-parallel over (stream, line, 8-sample group).  One workgroup renders kCompositeItemsPerBlock
// consecutive 8-sample groups of one stream's field (round 5: whole lines per workgroup left the last pass of
// an NTSC workgroup one eighth full -- 16 lines x 114 groups = 7.1 passes of 256 threads, against PAL's 8.9);
// the 3 KiB chroma-phase LUT lives in LDS; every thread produces one 16-byte (8-sample) store per pass, so
// line writes are fully coalesced.
//
// Two-frame horizontal scroll (_hscroll, video.cpp:1146-1154: the front frame from column h, then
// the other frame from column 0, each blit restarting its running luma average) and the 80 x 16
// time / progress-bar overlay (composite(), video.cpp:845-887, drawn on the black lines 2..17
// below the picture) are evaluated in the same pass.
//
// k_pdm restates pdm_second_order() (espflix.ino:73-107): the recurrence is serial per stream,
// so one lane owns one stream and walks its samples.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {


__device__ inline int luma_row_off(int y) { return (y >> 4) * kStripBytes + (y & 15) * kStride; }
__device__ inline int chroma_row_off(int plane, int c)
{
    return (c >> 3) * kStripBytes + ((c & 7) + (plane == 2 ? 8 : 0)) * kStride + EFX_FRAME_WIDTH;
}

__constant__ uint32_t c_dither[8] = {  // dither4x4, video.cpp:673-683: 4 lines x 2 frame phases
    0x00020301, 0x03010002, 0x02030100, 0x01000203, 0x03010002, 0x00020301, 0x01000203, 0x02030100};

}  // namespace

__global__ __launch_bounds__(256) void k_composite(const uint8_t* __restrict__ frames, const VideoTables* __restrict__ vt,
                                                   const VideoLineTemplates* __restrict__ lt, FieldArgs a,
                                                   uint16_t* __restrict__ out)
{
    const int first_stream = a.first_stream, ring_depth = a.ring_depth, frame_counter = a.frame_counter;
    __shared__ VideoTables v;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(vt);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&v);
        for (int i = threadIdx.x; i < (int)(sizeof(VideoTables) / 4); i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();

    const int groups = v.line_width / 8;  // 16-byte groups per line
    const int field_items = v.line_count * groups;
    const int blocks_per_field = (field_items + kCompositeItemsPerBlock - 1) / kCompositeItemsPerBlock;
    const int s = blockIdx.x / blocks_per_field;
    const int item0 = (blockIdx.x - s * blocks_per_field) * kCompositeItemsPerBlock;
    const uint32_t groups_magic = (0xFFFFFFFFu / (uint32_t)groups) + 1;  // item / groups = umulhi(item, magic): exact below 2^32 / groups
    const uint8_t* stream_frames = frames + (size_t)(first_stream + s) * ring_depth * kFrameBytes;
    // video_isr, video.cpp:1146-1154: a negative scroll shows the OTHER frame first
    int hscroll = a.hscroll, lead_slot = a.slot, trail_slot = a.other_slot;
    if (hscroll < 0) {
        hscroll += EFX_FRAME_WIDTH;
        lead_slot = a.other_slot;
        trail_slot = a.slot;
    }
    const int lead_groups = (EFX_FRAME_WIDTH - hscroll) / 4;  // 4-pixel groups taken from the leading frame
    const uint8_t* overlay = a.overlay ? a.overlay + (size_t)s * a.overlay_stride : nullptr;
    const int overlay_top = 32 + (v.pal ? 32 : 0) + 192 + 2;  // ptop, video.cpp:1182
    const int active_top = 32 + (v.pal ? 32 : 0);
    const int vsync_start = v.line_count - (v.pal ? 8 : 3);
    const int first_pix_group = (v.active_start + 16 + (v.pal ? 80 : 0)) / 8;

    for (int k = threadIdx.x; k < kCompositeItemsPerBlock; k += blockDim.x) {
        const int item = item0 + k;
        if (item >= field_items)
            break;
        const int i = (int)__umulhi((uint32_t)item, groups_magic), g = item - i * groups;
        uint4 o;
        const int pg = g - first_pix_group;
        const bool active = i >= active_top && i < active_top + 192;
        if (active && pg >= 0 && pg < EFX_FRAME_WIDTH / 4) {
            // blit(), video.cpp:690-804: 4 luma pixels -> 8 samples
            const int line = i - active_top;
            const int odd = line & 1;
            // source group: the leading frame from column hscroll, then the trailing frame from 0
            const bool lead = pg < lead_groups;
            const int sg = lead ? pg + hscroll / 4 : pg - lead_groups;
            const bool blit_start = lead ? pg == 0 : sg == 0;
            const uint8_t* frame = stream_frames + (size_t)(lead ? lead_slot : trail_slot) * kFrameBytes;
            const uint8_t* yrow = frame + luma_row_off(line);
            uint32_t y4 = *reinterpret_cast<const uint32_t*>(yrow + sg * 4);
            const int crow = line >> 1;
            const int coff = (sg >> 1) * 4;
            uint32_t u4 = *reinterpret_cast<const uint32_t*>(frame + chroma_row_off(1, crow) + coff);
            uint32_t v4 = *reinterpret_cast<const uint32_t*>(frame + chroma_row_off(2, crow) + coff);
            if (odd) {  // vertical chroma interpolation on odd lines (video.cpp:705-717)
                int n = crow + (line == 191 ? 0 : 1);
                uint32_t u2 = *reinterpret_cast<const uint32_t*>(frame + chroma_row_off(1, n) + coff);
                uint32_t v2 = *reinterpret_cast<const uint32_t*>(frame + chroma_row_off(2, n) + coff);
                u4 = ((u4 >> 1) & 0x7F7F7F7Fu) + ((u2 >> 1) & 0x7F7F7F7Fu);
                v4 = ((v4 >> 1) & 0x7F7F7F7Fu) + ((v2 >> 1) & 0x7F7F7F7Fu);
            }
            const uint32_t dither = c_dither[(line & 3) + ((frame_counter & 1) << 2)];
            // running luma average starts from the previous group's last pixel (0 where a blit starts)
            uint32_t lum = 0;
            if (!blit_start) {
                uint32_t yp = *reinterpret_cast<const uint32_t*>(yrow + sg * 4 - 4);
                lum = (((yp + dither) & 0xFCFCFCFCu) >> 2) >> 24;
            }
            uint32_t p0 = (y4 + dither) & 0xFCFCFCFCu;
            uint32_t p1 = ((p0 >> 1) + (p0 >> 9)) & 0xFCFCFCFCu;
            p0 >>= 2;
            p1 >>= 2;
            const uint32_t* tab_v = v.color_tab + (odd ? 512 : 256);
            const int sh = (sg & 1) * 16;
            uint32_t c = ((v.color_tab[(u4 >> sh) & 0xFF] + tab_v[(v4 >> sh) & 0xFF]) & 0xFCFCFCFCu) >> 2;
            lum = (((p0 & 0xFF) + lum) >> 1) & 0xFF;
            o.x = ((lum << 24) | ((p0 & 0xFF) << 8)) + c;
            o.y = ((p1 << 24) | (p0 & 0xFF00)) + (c << 8);
            c = ((v.color_tab[(u4 >> (sh + 8)) & 0xFF] + tab_v[(v4 >> (sh + 8)) & 0xFF]) & 0xFCFCFCFCu) >> 2;
            o.z = ((p1 << 16) | (p0 >> 8)) + c;
            o.w = (((p1 << 8) & 0xFF000000u) | (p0 >> 16)) + (c << 8);
        } else {
            // everything that is not picture: a 16-byte copy from the line's template
            int tpl;
            if (active || i < vsync_start)
                tpl = i & 1;  // template 0: _line_counter = i + 1 odd
            else if (!v.pal)
                tpl = 2;
            else {
                const int t = (0x00233000u >> ((i - 304) * 4)) & 0xF;  // _sync_type[8] = {0,0,0,3,3,2,0,0}
                tpl = t == 0 ? 3 : (t == 2 ? 4 : 5);
            }
            const uint4 tv = *reinterpret_cast<const uint4*>(&lt->tpl[tpl][g * 8]);
            uint32_t w[4] = {tv.x, tv.y, tv.z, tv.w};
            // composite(), video.cpp:845-887: overlay text = 80 bytes -> 160 samples starting 16
            // samples into the picture window; on overlay lines 3..8 a 240-step progress bar
            // follows after another 16 samples.  One overlay byte / bar step = one dword.
            const int ol = i - overlay_top;
            if (a.overlay_scale && ol >= 0 && ol < 16 && i < vsync_start) {
                const int d0 = (g - first_pix_group) * 4 - 8;  // dword index relative to the first text dword
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = d0 + k;
                    uint32_t p = 0;
                    bool hit = false;
                    if (d >= 0 && d < 80) {
                        p = v.black_level + (overlay ? overlay[ol * 80 + d] : 0) * a.overlay_scale;
                        hit = true;
                    } else if (ol >= 3 && ol <= 8 && d >= 88 && d < 88 + 240) {
                        const int step = (d - 88) & ~1;  // two dwords per step of 2
                        p = v.black_level + (step < a.overlay_progress ? (a.overlay_scale << 8) : (a.overlay_scale << 7));
                        hit = true;
                    }
                    if (hit)
                        w[k] = (p << 16) | p;
                }
            }
            o = make_uint4(w[0], w[1], w[2], w[3]);
        }
        // the field is written once and never re-read by this kernel: stream it past the caches
        uint16_t* dst = out + ((size_t)s * v.line_count + i) * v.line_width + g * 8;
        __builtin_nontemporal_store(o.x, reinterpret_cast<uint32_t*>(dst) + 0);
        __builtin_nontemporal_store(o.y, reinterpret_cast<uint32_t*>(dst) + 1);
        __builtin_nontemporal_store(o.z, reinterpret_cast<uint32_t*>(dst) + 2);
        __builtin_nontemporal_store(o.w, reinterpret_cast<uint32_t*>(dst) + 3);
    }
}

// pdm_second_order(), espflix.ino:73-107: two half-steps per PCM sample, 16 delta-sigma steps
// per half-step, one 16-bit word (MSB first) per half-step.
//
// The recurrence is serial, so the only costs worth fighting are the ones around it: PCM is read
// eight samples (16 bytes) at a time with the next group already in flight, the words of eight
// samples leave as two 16-byte stores, and a step is select arithmetic (no branch):
//   i1 += i0 - sign * a1 - (i2 >> 7);   i2 += i1 - sign * a2;   with sign = i2 >= 0 ? 1 : -1.
namespace {

// One delta-sigma step costs a wave EIGHT vector instructions (round 6; ten before): the sign of i2 as a mask (one shift),
// the feedback term (one shift), the two sign-dependent constants  i0 -+ a1  and  -+a2  each by ONE bit select on that mask
// (v_bitop3_b32, full rate; compare + three selects before), i1 in two additions, i2 in one three-operand addition, and the
// output bit folded into a running  nb = 2 nb + mask  (one v_add3_u32; the word is (nb - 1) & 0xFFFF).  At the batch the
// service runs (1024 streams = 16 waves on 1024 SIMDs) a wave is alone on its SIMD and issues one instruction per 4.4 cycles
// whatever it is (profiles/r6_valu_rates.md, `wave cyc`): the recurrence is serial per stream, so a stream-second costs
// 48 000 samples x 32 steps x instructions per step x 4.4 cycles -- the instruction COUNT of the step is the lever.
__device__ __forceinline__ uint32_t pdm_sel(uint32_t mask, uint32_t if_set, uint32_t if_clear)  // bit by bit: mask ? if_set : if_clear
{
    // (the builtin, not inline assembly: behind an asm statement the compiler's hazard recognizer puts an `s_nop`, and a wave
    // that is alone on its SIMD pays 4.4 cycles for that like for any other instruction)
    return __builtin_amdgcn_bitop3_b32(mask, if_set, if_clear, 0xCA);
}

__device__ __forceinline__ uint32_t pdm_sample(int32_t pcm, uint32_t& i0, uint32_t& i1, uint32_t& i2)
{
    const uint32_t a1 = 38973;  // int32(0x7FFF * 1.18940)
    const uint32_t a2 = 69577;  // int(0x7FFF * 2.12340)
    const uint32_t x = (uint32_t)(pcm * 2);
    uint32_t word = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        i0 = (uint32_t)(((int32_t)(i0 + x)) >> 1);
        const uint32_t c_pos = i0 - a1, c_neg = i0 + a1;  // what i1 gains besides the feedback, by the sign of i2
        uint32_t nb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t neg = (uint32_t)((int32_t)i2 >> 31);  // all ones: i2 < 0
            asm("" : "+v"(neg));                           // (a mask, not a condition: the compiler would rebuild compare + selects)
            const uint32_t fb = (uint32_t)(((int32_t)i2) >> 7);
            i1 = (i1 - fb) + pdm_sel(neg, c_neg, c_pos);
            i2 = i2 + i1 + pdm_sel(neg, a2, 0u - a2);
            nb = nb + nb + neg;
            asm("" : "+v"(nb));  // (one v_lshl_add_u32 per step: left alone the compiler re-associates two steps into three instructions)
        }
        word |= ((nb - 1u) & 0xFFFF) << (16 * half);  // consecutive uint16 words, little endian
    }
    return word;
}

}  // namespace

__global__ void k_pdm(const int16_t* __restrict__ pcm, int n_streams, int n_samples, int32_t* __restrict__ state,
                      uint16_t* __restrict__ dst)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams)
        return;
    uint32_t i0 = (uint32_t)state[s * 3 + 0], i1 = (uint32_t)state[s * 3 + 1], i2 = (uint32_t)state[s * 3 + 2];
    const int16_t* src = pcm + (size_t)s * n_samples;
    uint32_t* out = reinterpret_cast<uint32_t*>(dst + (size_t)s * 2 * n_samples);
    int n = 0;
    // groups of eight samples when this stream's PCM and output are 16-byte aligned
    if ((((uintptr_t)src | (uintptr_t)out) & 15) == 0 && n_samples >= 8) {
        const int groups = n_samples / 8;
        const uint4* src4 = reinterpret_cast<const uint4*>(src);
        uint4* out4 = reinterpret_cast<uint4*>(out);
        uint4 cur = src4[0];
        for (int g = 0; g < groups; g++) {
            const uint4 nxt = src4[g + 1 < groups ? g + 1 : g];  // in flight while this group is modulated
            const uint32_t in[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                w[2 * k] = pdm_sample((int16_t)(in[k] & 0xFFFF), i0, i1, i2);
                w[2 * k + 1] = pdm_sample((int16_t)(in[k] >> 16), i0, i1, i2);
            }
            out4[2 * g] = make_uint4(w[0], w[1], w[2], w[3]);
            out4[2 * g + 1] = make_uint4(w[4], w[5], w[6], w[7]);
            cur = nxt;
        }
        n = groups * 8;
    }
    for (; n < n_samples; n++)
        out[n] = pdm_sample(src[n], i0, i1, i2);
    state[s * 3 + 0] = (int32_t)i0;
    state[s * 3 + 1] = (int32_t)i1;
    state[s * 3 + 2] = (int32_t)i2;
}

}  // namespace efx
