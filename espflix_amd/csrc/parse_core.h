// parse_core.h -- the per-slice VLC parser of k_parse as lane-level code that also compiles for the host.
//
// Restates MpegDecoder::slice() (reference src/player.cpp:1251-1316), motion_vector(s) (891-920) and
// the entropy-decode half of block() (999-1107) as a TWO-STATE MACHINE per slice:
//
//   kLaneCoef     the lane is inside a block's run/level list: one trip = one DCT symbol through the
//                 flat table (plus the end_of_block that follows it, if it does);
//   kLaneService  the lane is at a block or macroblock boundary and waits for `service()`: close the
//                 block, write the macroblock record, decode the next macroblock header (address
//                 increment, type, quantiser, motion vectors, coded block pattern), start the next
//                 block (intra: DC size + differential).
//
// k_parse runs 64 slices per wave.  The coefficient step is short and runs every trip; the service
// step is long and runs only when enough lanes wait for it (or none can go on), so its cost is shared
// by many lanes instead of being paid by the whole wave for every lane that reaches a boundary.
// tests/parse_harness.cpp compiles this file for the host and checks every record and coefficient
// entry against a trace of the test oracle.
#pragma once
#include <cstdint>

#include "efx.h"
#include "efx_internal.h"

#if defined(__HIPCC__)
#define EFX_HD __host__ __device__ __forceinline__
#else
#define EFX_HD inline
#endif

namespace efx {

enum : uint32_t { kLaneCoef = 0, kLaneService = 1, kLaneDone = 2 };

EFX_HD int efx_clz32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)v);
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

// constants of one slice
struct SliceParams {
    uint32_t coef_last;  // last coefficient slot of the slice's region
    uint32_t i_picture;  // macroblock_type book: table B-2a (1) or B-2b (0)
    uint32_t full_pel;
    uint32_t r_size;     // forward_f_code - 1
    uint32_t rec_flags;  // 0x80 when the picture uses loaded quantiser matrices
    uint32_t epoch;
};

// DC size + differential of an intra block, player.cpp:1010-1068 (tables B-5a / B-5b): at most
// 10 + 11 bits, all inside one window.  Updates the predictor, returns the DC value and the bits used.
EFX_HD int decode_dc(uint32_t win, int blk, int& dc_y, int& dc_cr, int& dc_cb, uint32_t& used)
{
    int size, len, pred;
    if (blk < 4) {
        uint32_t pb = win >> 23;
        int ones = efx_clz32(~(pb << 23));
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
        uint32_t pb = win >> 22;
        int ones = efx_clz32(~(pb << 22));
        if (ones == 0) {
            size = (int)((pb >> 8) & 1);
            len = 2;
        } else {
            size = ones + 1;
            len = size < 10 ? size : 10;
        }
        pred = (blk == 4) ? dc_cr : dc_cb;
    }
    if (size) {
        int delta = (int)((win << len) >> (32 - size));
        len += size;
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
    used = (uint32_t)len;
    return pred;
}

// One slice.  BR supplies window() -- the next 32 bits of the slice, MSB first -- and advance(n).
template <class BR, class Tab, bool kAllIntra>
struct SliceParser {
    uint32_t st;
    // slice() state, player.cpp:1255-1263
    int mb_addr;
    int dc_y, dc_cr, dc_cb;
    int mv_h, mv_v;
    int qscale;
    // macroblock in progress
    uint32_t cbp_left;  // coded blocks not yet started (bit 5 = block 0)
    int blk;            // block in progress, -1 = none
    int n;              // scan position of the next coefficient
    uint32_t first;     // the next symbol is the first of a non-intra block ("1s" = (0, +-1))
    uint32_t intra;
    uint32_t coef_idx, blk_start, mb_coef_base;
    uint32_t cnt_lo, cnt_hi;  // entries per block, one byte each: blocks 0-3 / 4-5
    uint32_t rec_flags, rec_mv;
    uint32_t mb_open;   // a macroblock record is pending
    uint32_t first_mb;  // the next macroblock is the slice's first (forced to column 0)
    uint32_t in_mba;    // macroblock_address_increment in progress (stuffing / escape seen)
    uint32_t esc_seen;
    int inc_acc;
    uint32_t bad, dropped;
    uint32_t status, n_coefs, n_mbs;

    // slice header, player.cpp:1255-1263: quantiser_scale, extra_information_slice
    EFX_HD void begin(BR& br, int code, uint32_t coef_base)
    {
        mb_addr = (code - 1) * kMbW - 1;  // mb_y = code - 2, mb_x = mb_width - 1
        dc_y = dc_cr = dc_cb = 128;
        mv_h = mv_v = 0;
        cbp_left = 0;
        blk = -1;
        n = 0;
        first = intra = 0;
        coef_idx = blk_start = mb_coef_base = coef_base;
        cnt_lo = cnt_hi = rec_flags = rec_mv = 0;
        mb_open = 0;
        first_mb = 1;
        in_mba = esc_seen = 0;
        inc_acc = 0;
        bad = dropped = 0;
        status = n_coefs = n_mbs = 0;
        uint32_t win = br.window();
        qscale = (int)(win >> 27);
        // extra_bit_slice = 1: skip it and 8 bits of information; two such groups fit the window, a
        // longer run is finished by service() (in_mba = 2)
        uint32_t used = 5;
        int groups = 0;
        while (((win << used) >> 31) && groups < 2) {
            used += 9;
            groups++;
        }
        if ((win << used) >> 31) {
            in_mba = 2;  // more extra information follows
            br.advance(used);
        } else
            br.advance(used + 1);
        st = kLaneService;
    }

    EFX_HD void write_record(MbRec* recs, const SliceParams& sp)
    {
        // MbRec: coef_base | cnt[0..3] | cnt[4] cnt[5] flags epoch | mvx mvy
        uint32_t w[4] = {mb_coef_base, cnt_lo, cnt_hi | (rec_flags << 16) | ((sp.epoch & 0xFF) << 24), rec_mv};
#if defined(__HIP_DEVICE_COMPILE__)
        *reinterpret_cast<uint4*>(&recs[mb_addr]) = make_uint4(w[0], w[1], w[2], w[3]);
#else
        uint32_t* d = reinterpret_cast<uint32_t*>(&recs[mb_addr]);
        d[0] = w[0], d[1] = w[1], d[2] = w[2], d[3] = w[3];
#endif
        n_mbs++;
    }

    // motion_vector(), player.cpp:891-910.  At most 11 + 6 bits: one window.
    EFX_HD int motion(BR& br, const Tab& t, int pred, int r_size, bool& ok)
    {
        uint32_t win = br.window();
        uint32_t e = t.motion[win >> 21];
        uint32_t len = e & 15;
        if (!len) {
            ok = false;
            return pred;
        }
        int code = (int)(e >> 4) - 16;
        int d = code;
        if (code != 0 && r_size != 0) {
            int a = code < 0 ? -code : code;
            d = ((a - 1) << r_size) + (int)((win << len) >> (32 - r_size)) + 1;
            len += r_size;
            if (code < 0)
                d = -d;
        }
        br.advance(len);
        int scale = 1 << r_size;
        int m = pred + d;
        if (m > (scale << 4) - 1)
            m -= scale << 5;
        else if (m < -(scale << 4))
            m += scale << 5;
        return m;
    }

    // ---- one DCT symbol (player.cpp:1070-1107); call for lanes in kLaneCoef ----------------------------
    EFX_HD void coef_step(BR& br, const Tab& t, uint32_t* coefs, const SliceParams& sp)
    {
        const uint32_t win = br.window();
        const uint32_t pk = win >> 16;
        uint32_t ent = (pk >= 0x0400) ? t.dct_hi[pk >> 8] : t.dct_lo[pk & 0x3FF];
        if (!kAllIntra) {
            // first coefficient of a non-intra block: "1s" = (0, +-1); end_of_block cannot come first
            if (first && (win >> 31))
                ent = 2u | (1u << 10);
            first = 0;
        }
        // entry: bits consumed (code + sign; 2 for end_of_block; 20 for an escape with an 8-bit level; 0 =
        // invalid) | run << 5 | level << 10 (0 = escape, 63 = end_of_block)
        const uint32_t len_f = ent & 31, run_f = (ent >> 5) & 31, lev_f = ent >> 10;
        const bool invalid = len_f == 0;
        bool eob = lev_f == 63;
        const bool esc = lev_f == 0;  // escape: 6-bit run, 8- or 16-bit level (player.cpp:1092-1099)
        const int lvl_n = ((win << ((len_f - 1) & 31)) >> 31) ? -(int)lev_f : (int)lev_f;  // sign = last bit of the code
        const uint32_t lv8 = (win << 12) >> 24, ext = (win << 20) >> 24;
        const bool two = (lv8 & 0x7F) == 0;
        const int lvl_e = two ? (lv8 ? (int)ext - 256 : (int)ext) : (lv8 > 128 ? (int)lv8 - 256 : (int)lv8);
        const int level = esc ? lvl_e : lvl_n;
        const uint32_t run = esc ? (win << 6) >> 26 : run_f;
        uint32_t len = (esc && two) ? 28u : len_f;
        const int n_new = n + (int)run;
        const bool drop = !eob && !invalid && n_new >= 64;  // player.cpp:1106-1107: block abandoned
        const bool keep = !(eob || invalid || drop);
        if (keep) {
            const uint32_t at = coef_idx < sp.coef_last ? coef_idx : sp.coef_last;
            coefs[at] = ((uint32_t)level << 6) | (uint32_t)(n_new & 63);
            coef_idx++;
            // the end_of_block that follows this symbol is taken in the same trip
            if (((win << len) >> 30) == 2) {
                len += 2;
                eob = true;
            }
        }
        n = n_new + 1;
        br.advance(len);
        bad |= invalid ? 1u : 0u;
        dropped |= drop ? 1u : 0u;
        if (eob || invalid || drop)
            st = kLaneService;
    }

    // ---- block / macroblock boundary work; call for lanes in kLaneService ------------------------------
    EFX_HD void service(BR& br, const Tab& t, uint32_t* coefs, MbRec* recs, const SliceParams& sp)
    {
        // A. close the block that ended
        if (blk >= 0) {
            if (dropped) {
                status |= EFX_STREAM_COEF_OVERRUN;
                coef_idx = blk_start;  // forget the partial block
                dropped = 0;
            } else if (!bad) {
                const uint32_t c = coef_idx - blk_start;
                n_coefs += c;
                if (blk < 4)
                    cnt_lo |= c << (8 * blk);
                else
                    cnt_hi |= c << (8 * (blk - 4));
            }
            blk = -1;
            if (bad) {
                write_record(recs, sp);
                status |= EFX_STREAM_BAD_VLC;
                st = kLaneDone;
                return;
            }
        }
        // B. macroblock boundary: record of the finished macroblock, header of the next
        if (cbp_left == 0) {
            if (mb_open) {
                write_record(recs, sp);
                mb_open = 0;
                if (coef_idx > sp.coef_last) {
                    status |= EFX_STREAM_BAD_VLC;  // ran past this slice's bytes without finding its end
                    st = kLaneDone;
                    return;
                }
            }
            uint32_t win = br.window();
            if (in_mba == 2) {  // slice header: further extra_information_slice groups
                uint32_t used = 0;
                int groups = 0;
                while (((win << used) >> 31) && groups < 3) {
                    used += 9;
                    groups++;
                }
                if ((win << used) >> 31) {
                    br.advance(used);
                    return;
                }
                br.advance(used + 1);
                in_mba = 0;
                return;
            }
            if (!in_mba && (win >> 9) == 0) {  // slice_done(): 23 zero bits, player.cpp:1238-1249
                st = kLaneDone;
                return;
            }
            // macroblock_address_increment with stuffing (34) and escape (35), player.cpp:1267-1275
            const uint32_t e = t.mba[win >> 21];
            const uint32_t l0 = e & 15;
            const int v = (int)(e >> 4);
            if (!l0) {
                status |= EFX_STREAM_BAD_VLC;
                st = kLaneDone;
                return;
            }
            if ((v == 34 && !esc_seen) || v == 35) {
                if (v == 35) {
                    inc_acc += 33;
                    esc_seen = 1;
                }
                in_mba = 1;
                br.advance(l0);
                return;  // the next code word is read by the next service pass
            }
            int inc = inc_acc + v;
            inc_acc = 0;
            in_mba = esc_seen = 0;

            if (first_mb) {
                mb_addr += 1;  // inc_mb() ignores its argument: first macroblock -> column 0 (player.cpp:823-833,1277)
                first_mb = 0;
            } else {
                if (inc > 1) {
                    dc_y = dc_cr = dc_cb = 128;  // reset_predictors(), player.cpp:1280-1281
                    mv_h = mv_v = 0;
                }
                while (inc > 1 && mb_addr + 1 < kMbCount) {  // skipped macroblocks copy the reference (1283-1288)
                    mb_addr++;
                    mb_coef_base = coef_idx;
                    cnt_lo = cnt_hi = 0;
                    rec_flags = 2;
                    rec_mv = 0;
                    write_record(recs, sp);
                    inc--;
                }
                mb_addr++;
            }
            if (mb_addr >= kMbCount) {
                status |= EFX_STREAM_MB_OVERRUN;
                st = kLaneDone;
                return;
            }

            // macroblock_type (+ quantiser_scale), player.cpp:1292-1296: at most 6 + 5 bits, still inside
            // the window that held the address increment (11 bits at most)
            const uint32_t w2 = win << l0;
            int type;
            uint32_t used;
            if (kAllIntra || sp.i_picture) {
                if (w2 >> 31) {
                    type = 1;
                    used = 1;
                } else if ((w2 >> 30) == 1) {
                    type = 17;
                    used = 2;
                } else {
                    status |= EFX_STREAM_BAD_VLC;
                    st = kLaneDone;
                    return;
                }
            } else {
                const uint32_t tt = t.type_p[w2 >> 26];
                if (!(tt & 7)) {
                    status |= EFX_STREAM_BAD_VLC;
                    st = kLaneDone;
                    return;
                }
                used = tt & 7;
                type = (int)(tt >> 3);
            }
            intra = (uint32_t)type & 1;
            if (type & 0x10) {
                qscale = (int)((w2 << used) >> 27);
                used += 5;
            }
            br.advance(l0 + used);

            mb_coef_base = coef_idx;
            cnt_lo = cnt_hi = 0;
            if (intra) {
                mv_h = mv_v = 0;  // player.cpp:1300
            } else {
                dc_y = dc_cr = dc_cb = 128;  // player.cpp:1302
                if (type & 0x08) {
                    bool ok = true;
                    mv_h = motion(br, t, mv_h, (int)sp.r_size, ok);
                    mv_v = motion(br, t, mv_v, (int)sp.r_size, ok);
                    if (!ok) {
                        status |= EFX_STREAM_BAD_VLC;
                        st = kLaneDone;
                        return;
                    }
                } else
                    mv_h = mv_v = 0;
            }
            // predict(), player.cpp:878-881: full-pel vectors are doubled
            rec_mv = ((uint32_t)(sp.full_pel ? mv_h << 1 : mv_h) & 0xFFFF) | ((uint32_t)(sp.full_pel ? mv_v << 1 : mv_v) << 16);
            rec_flags = (intra ? 1u : 0u) | ((uint32_t)qscale << 2) | sp.rec_flags;
            uint32_t cbp = intra ? 63u : 0u;
            if (type & 0x02) {
                const uint32_t c = t.cbp[br.window() >> 23];
                if (!(c & 15)) {
                    status |= EFX_STREAM_BAD_VLC;
                    st = kLaneDone;
                    return;
                }
                br.advance(c & 15);
                cbp = c >> 4;
            }
            cbp_left = cbp;
            mb_open = 1;
            if (!cbp)
                return;  // motion only: the next service pass writes the record and reads the next header
        }
        // C. start the next coded block
        blk = efx_clz32(cbp_left) - 26;
        cbp_left &= ~(0x20u >> blk);
        blk_start = coef_idx;
        if (intra) {
            const uint32_t win = br.window();
            uint32_t used;
            const int dc = decode_dc(win, blk, dc_y, dc_cr, dc_cb, used);
            const uint32_t at = coef_idx < sp.coef_last ? coef_idx : sp.coef_last;
            coefs[at] = (uint32_t)dc << 6;
            coef_idx++;
            n = 1;
            first = 0;
            if (((win << used) >> 30) == 2) {
                br.advance(used + 2);  // DC only: the block is complete, the next service pass closes it
                return;
            }
            br.advance(used);
        } else {
            n = 0;
            first = 1;
        }
        st = kLaneCoef;
    }
};

}  // namespace efx
