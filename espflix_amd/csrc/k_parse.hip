// k_parse.hip -- slice / macroblock / block VLC parse + dequantisation (gfx950).
//
// ONE LANE PER SLICE.  Restates MpegDecoder::slice() (reference src/player.cpp:1251-1316),
// motion_vector(s) (891-920) and the entropy-decode + reconstruction half of block()
// (999-1122) with flat look-up tables staged in LDS instead of the reference's bit-serial tree
// walk (516-530) and prefix-class DCT decoder (548-644).  Slices are independent: every
// predictor is reset at the slice header (1260), so a batch of B streams x P pictures x S
// slices gives B*P*S parallel lanes; descriptors arrive in picture-major order so the 64
// lanes of a wave hold slices of the same picture type.
//
// Output per macroblock: a 16-byte MbRec (type, motion vector, per-block coefficient counts)
// and the macroblock's coefficients as compact 32-bit entries
//   entry = (dequantised value * IDCT pre-multiplier) << 6 | raster position,
// i.e. exactly the non-zero b[zz] of player.cpp:1121; the dense 64-int block never exists in
// memory.  k_recon turns these into pixels.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {

struct BitReader {
    const uint32_t* p;  // next aligned dword
    uint64_t w;         // MSB-aligned window
    int cnt;            // valid bits in w

    __device__ inline void refill()
    {
        if (cnt <= 32) {
            uint32_t d = __builtin_bswap32(*p++);
            w |= (uint64_t)d << (32 - cnt);
            cnt += 32;
        }
    }
    __device__ inline void init(const uint8_t* ptr)
    {
        uintptr_t a = (uintptr_t)ptr;
        int mis = (int)(a & 3);
        p = (const uint32_t*)(a - mis);
        w = 0;
        cnt = 0;
        refill();
        w <<= mis * 8;
        cnt -= mis * 8;
        refill();
    }
    __device__ inline uint32_t peek(int n) const { return (uint32_t)(w >> (64 - n)); }  // 1 <= n <= 32
    __device__ inline void skip(int n)
    {
        w <<= n;
        cnt -= n;
    }
    __device__ inline uint32_t get(int n)
    {
        uint32_t v = peek(n);
        skip(n);
        return v;
    }
};

struct SharedTables {
    ParseTables t;
};

// motion_vector(), player.cpp:891-910
__device__ inline int decode_motion(BitReader& br, const uint16_t* tab, int pred, int r_size, bool& ok)
{
    br.refill();
    uint32_t e = tab[br.peek(11)];
    int len = e & 15;
    if (!len) {
        ok = false;
        return pred;
    }
    br.skip(len);
    int code = (int)(e >> 4) - 16;
    int d = code;
    if (code != 0 && r_size != 0) {
        int a = code < 0 ? -code : code;
        d = ((a - 1) << r_size) + (int)br.get(r_size) + 1;
        if (code < 0)
            d = -d;
    }
    int scale = 1 << r_size;
    int m = pred + d;
    if (m > (scale << 4) - 1)
        m -= scale << 5;
    else if (m < -(scale << 4))
        m += scale << 5;
    return m;
}

}  // namespace

__global__ __launch_bounds__(256) void k_parse(const uint8_t* __restrict__ es, const SliceDesc* __restrict__ descs,
                                               DecodeCounters* __restrict__ counters,
                                               const ParseTables* __restrict__ gtab,
                                               const uint32_t* __restrict__ qtab_custom, MbRec* __restrict__ mbrecs,
                                               uint32_t* __restrict__ coefs, uint32_t* __restrict__ status,
                                               int max_pictures, int epoch)
{
    __shared__ SharedTables sh;
    {
        // stage the look-up tables: sizeof(ParseTables) is a multiple of 4
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gtab);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.t);
        for (int i = threadIdx.x; i < (int)(sizeof(ParseTables) / 4); i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= counters->total_slices)
        return;

    const SliceDesc d = descs[gid];
    const uint32_t pic = d.pic_code_flags & 0xFF;
    const int code = (d.pic_code_flags >> 8) & 0xFF;
    const bool i_picture = ((d.pic_code_flags >> 16) & 3) == 1;
    const int full_pel = (d.pic_code_flags >> 18) & 1;
    const int r_size = (d.pic_code_flags >> 19) & 7;
    const bool custom_q = (d.pic_code_flags >> 22) & 1;
    const uint32_t* qtab = custom_q ? qtab_custom + ((size_t)d.stream * max_pictures + pic) * 64 : nullptr;

    MbRec* recs = mbrecs + ((size_t)d.stream * max_pictures + pic) * kMbCount;
    uint32_t coef_idx = d.es_off * kCoefsPerEsByte;
    const uint32_t coef_end = (d.es_off + d.es_len) * kCoefsPerEsByte;

    BitReader br;
    br.init(es + d.es_off);

    uint32_t st = 0;
    uint32_t n_coefs = 0, n_mbs = 0;

    // slice header, player.cpp:1255-1263
    int mb_addr = (code - 1) * kMbW - 1;  // mb_y = code-2, mb_x = mb_width-1
    int dc_y = 128, dc_cr = 128, dc_cb = 128;
    int mv_h = 0, mv_v = 0;
    int qscale = (int)br.get(5);
    br.refill();
    while (br.get(1)) {
        br.skip(8);
        br.refill();
    }

    for (int mb = 0;; mb++) {
        br.refill();
        if (br.peek(23) == 0)  // slice_done(), player.cpp:1238-1249
            break;

        // macroblock_address_increment with stuffing (34) and escape (35), player.cpp:1267-1275
        int inc = 0;
        uint32_t e;
        int v;
        bool ok = true;
        do {
            br.refill();
            e = sh.t.mba[br.peek(11)];
            if (!(e & 15)) {
                ok = false;
                break;
            }
            br.skip(e & 15);
            v = (int)(e >> 4);
        } while (v == 34);
        while (ok && v == 35) {
            inc += 33;
            br.refill();
            e = sh.t.mba[br.peek(11)];
            if (!(e & 15)) {
                ok = false;
                break;
            }
            br.skip(e & 15);
            v = (int)(e >> 4);
        }
        if (!ok) {
            st |= EFX_STREAM_BAD_VLC;
            break;
        }
        inc += v;

        if (mb == 0) {
            mb_addr += 1;  // inc_mb() ignores its argument: first macroblock -> column 0 (player.cpp:823-833,1277)
        } else {
            if (inc > 1) {
                dc_y = dc_cr = dc_cb = 128;  // reset_predictors(), player.cpp:1280-1281
                mv_h = mv_v = 0;
            }
            while (inc > 1 && mb_addr + 1 < kMbCount) {  // skipped macroblocks copy the reference (1283-1288)
                mb_addr++;
                MbRec r;
                r.coef_base = coef_idx;
                for (int k = 0; k < 6; k++)
                    r.cnt[k] = 0;
                r.flags = 2;
                r.epoch = (uint8_t)epoch;
                r.mvx = r.mvy = 0;
                recs[mb_addr] = r;
                n_mbs++;
                inc--;
            }
            mb_addr++;
        }
        if (mb_addr >= kMbCount) {
            st |= EFX_STREAM_MB_OVERRUN;
            break;
        }

        // macroblock_type, player.cpp:1292-1296
        br.refill();
        int type;
        if (i_picture) {
            uint32_t pk = br.peek(2);
            if (pk & 2) {
                type = 1;
                br.skip(1);
            } else if (pk == 1) {
                type = 17;
                br.skip(2);
            } else {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
        } else {
            uint32_t t = sh.t.type_p[br.peek(6)];
            if (!(t & 7)) {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
            br.skip(t & 7);
            type = (int)(t >> 3);
        }
        const bool intra = type & 1;
        if (type & 0x10)
            qscale = (int)br.get(5);

        MbRec rec;
        rec.coef_base = coef_idx;
        rec.epoch = (uint8_t)epoch;
        rec.flags = intra ? 1 : 0;
        if (intra) {
            mv_h = mv_v = 0;  // player.cpp:1300
        } else {
            dc_y = dc_cr = dc_cb = 128;  // player.cpp:1302
            if (type & 0x08) {
                mv_h = decode_motion(br, sh.t.motion, mv_h, r_size, ok);
                mv_v = decode_motion(br, sh.t.motion, mv_v, r_size, ok);
                if (!ok) {
                    st |= EFX_STREAM_BAD_VLC;
                    break;
                }
            } else
                mv_h = mv_v = 0;
        }
        rec.mvx = (int16_t)(full_pel ? mv_h << 1 : mv_h);  // predict(), player.cpp:878-881
        rec.mvy = (int16_t)(full_pel ? mv_v << 1 : mv_v);

        int cbp = intra ? 63 : 0;
        if (type & 0x02) {
            br.refill();
            uint32_t c = sh.t.cbp[br.peek(9)];
            if (!(c & 15)) {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
            br.skip(c & 15);
            cbp = (int)(c >> 4);
        }

        bool bad = false;
        for (int blk = 0; blk < 6; blk++) {
            rec.cnt[blk] = 0;
            if (!(cbp & (0x20 >> blk)))
                continue;
            const uint32_t blk_start = coef_idx;
            int n = 0;
            br.refill();
            if (intra) {
                // DC size + differential, player.cpp:1010-1068 (table B-5a / B-5b)
                int size, len, pred;
                if (blk < 4) {
                    uint32_t pb = br.peek(9);
                    int ones = __clz((int)~(pb << 23));
                    if (ones == 0) {
                        size = 1 + (int)((pb >> 7) & 1);
                        len = 2;
                    } else if (ones == 1) {
                        size = (pb & 0x40) ? 3 : 0;
                        len = 3;
                    } else {
                        size = ones + 2;
                        len = ones + 1;
                    }
                    pred = dc_y;
                } else {
                    uint32_t pb = br.peek(10);
                    int ones = __clz((int)~(pb << 22));
                    if (ones == 0) {
                        size = (int)((pb >> 8) & 1);
                        len = 2;
                    } else {
                        size = ones + 1;
                        len = size < 10 ? size : 10;
                    }
                    pred = (blk == 4) ? dc_cr : dc_cb;
                }
                br.skip(len);
                if (size) {
                    br.refill();
                    int delta = (int)br.get(size);
                    if (delta & (1 << (size - 1)))
                        pred += delta;
                    else
                        pred += (int)((~0u << size) | (uint32_t)(delta + 1));
                    if (blk == 4)
                        dc_cr = pred;
                    else if (blk == 5)
                        dc_cb = pred;
                    else
                        dc_y = pred;
                }
                if (coef_idx < coef_end)
                    coefs[coef_idx] = ((uint32_t)pred << 8) << 6;  // b[0] = dc << 8, zz = 0
                coef_idx++;
                n = 1;
            }

            bool dropped = false;
            for (;;) {  // run/level pairs, player.cpp:1070-1122
                br.refill();
                uint32_t pk = br.peek(16);
                int run, level;
                if (pk & 0x8000) {
                    if (n && !(pk & 0x4000)) {  // "10": end_of_block (not possible as first code)
                        br.skip(2);
                        break;
                    }
                    // "1s" as first coefficient of a non-intra block, "11s" otherwise: (0, +-1)
                    br.skip(n ? 2 : 1);
                    run = 0;
                    level = br.get(1) ? -1 : 1;
                } else {
                    uint32_t ent = (pk >= 0x0400) ? sh.t.dct_hi[pk >> 8] : sh.t.dct_lo[pk & 0x3FF];
                    int len = ent & 31;
                    if (!len) {
                        bad = true;
                        break;
                    }
                    br.skip(len);
                    run = (ent >> 5) & 31;
                    level = (int)(ent >> 10);
                    if (level == 0) {  // escape: 6-bit run, 8- or 16-bit level (player.cpp:1092-1099)
                        run = (int)br.get(6);
                        br.refill();
                        level = (int)br.get(8);
                        if (level == 0)
                            level = (int)br.get(8);
                        else if (level == 128)
                            level = (int)br.get(8) - 256;
                        else if (level > 128)
                            level -= 256;
                    } else if (br.get(1))
                        level = -level;
                }
                n += run;
                if (n >= 64) {  // player.cpp:1106-1107: the block is abandoned, nothing is stored
                    dropped = true;
                    break;
                }
                uint32_t t = qtab ? qtab[n] : sh.t.scan[n];
                n++;
                int q = intra ? (int)((t >> 16) & 0xFF) : (int)(t >> 24);
                // reconstruction, player.cpp:1110-1121
                int val = level << 1;
                if (!intra)
                    val += (val < 0) ? -1 : 1;
                val = val * qscale * q;
                val = (val + ((val >> 31) & 15)) >> 4;  // division by 16 truncating toward zero
                if ((val & 1) == 0)
                    val -= (val > 0) ? 1 : -1;
                val = val > 2047 ? 2047 : (val < -2048 ? -2048 : val);
                val *= (int)((t >> 8) & 0xFF);
                if (coef_idx < coef_end)
                    coefs[coef_idx] = ((uint32_t)val << 6) | (t & 0x3F);
                coef_idx++;
            }
            if (bad)
                break;
            if (dropped) {
                st |= EFX_STREAM_COEF_OVERRUN;
                coef_idx = blk_start;  // forget the partial block
                rec.flags |= (uint8_t)(4u << blk);
            } else {
                rec.cnt[blk] = (uint8_t)(coef_idx - blk_start);
                n_coefs += coef_idx - blk_start;
            }
        }
        recs[mb_addr] = rec;
        n_mbs++;
        if (bad) {
            st |= EFX_STREAM_BAD_VLC;
            break;
        }
    }
    if (coef_idx > coef_end)
        st |= EFX_STREAM_BAD_VLC;
    if (st)
        atomicOr(&status[d.stream], st);
    atomicAdd(&counters->coefficients, (unsigned long long)n_coefs);
    atomicAdd(&counters->macroblocks, (unsigned long long)n_mbs);
}

}  // namespace efx
