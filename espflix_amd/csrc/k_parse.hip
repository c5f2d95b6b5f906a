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
//   entry = signed level << 6 | scan position      (intra DC: DC value << 6 | 0),
// the dense 64-int block never exists in memory.  Dequantisation (player.cpp:1110-1121) happens
// in k_recon, where one lane handles one coefficient: in this kernel only ~1/4 of the lanes do
// useful work per instruction (lock-step slices), so every instruction moved out of the symbol
// loop is paid for four times over.
//
// The kernel is bound by the serial symbol chain of the longest slices (I pictures), i.e. by
// instructions per symbol, so the symbol loop is kept minimal: the bit reader is a bit POSITION
// into a per-lane LDS ring (one 32-bit window per symbol, no refill state), every DCT code
// including "10"/"11s" resolves through one table look-up, each symbol is stored one iteration
// late (in the shadow of the next look-up), and global memory is touched only by that store and
// by wave-synchronous ring top-ups.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {


namespace {

constexpr int kRingDwords = 16;  // per-lane bitstream ring in LDS (64 bytes)
constexpr int kRingLow = 8;      // top up when any lane of the wave has fewer dwords than this ahead

// Bit reader.  Each lane owns a ring of kRingDwords big-endian dwords of ITS slice in LDS, laid
// out ring[k][lane] (a wave's accesses hit 64 different banks).  Global memory is read in
// WAVE-SYNCHRONOUS top-ups: when any lane runs low every lane refills its ring to the brim with
// independent loads, so the HBM/L2 latency is paid once per ~100 symbols.
struct BitReader {
    const uint32_t* gp;  // dword 0 of this lane's slice (aligned down); global address space
    uint32_t* ring;      // &ring[0][lane]; dword k lives at ring[(k % kRingDwords) * 64]
    uint32_t pos;        // bit position relative to gp
    uint32_t wr;         // dwords copied global -> ring so far

#ifdef EFX_DEBUG_WAVES
    unsigned long long dbg_topup = 0;
#endif
    __device__ inline void topup()
    {
        if (__any((int)(wr - (pos >> 5)) < kRingLow)) {
#ifdef EFX_DEBUG_WAVES
            const unsigned long long t0 = __builtin_readcyclecounter();
#endif
            const uint32_t n = kRingDwords - (wr - (pos >> 5));
            for (uint32_t base = 0; base < (uint32_t)kRingDwords; base += 8) {
                if (!__any(base < n))
                    break;
                // eight independent loads in flight, then eight UNCONDITIONAL ring writes (lanes
                // that need fewer dwords aim the surplus at a scratch row): no load is left
                // pending on any path, so the symbol loop carries no vmcnt wait
                uint32_t v[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++)
                    v[j] = gp[wr + ((base + j < n) ? base + j : 0)];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    uint32_t slot = (base + j < n) ? (wr + base + j) % kRingDwords : (uint32_t)kRingDwords;
                    ring[slot * 64] = __builtin_bswap32(v[j]);
                }
            }
            wr += n;
#ifdef EFX_DEBUG_WAVES
            dbg_topup += __builtin_readcyclecounter() - t0;
#endif
        }
    }
    __device__ inline void init(const uint8_t* __restrict__ es, uint32_t off, uint32_t* ring_lane)
    {
        uint32_t mis = off & 3;  // the ES buffer itself is 256-byte aligned
        gp = reinterpret_cast<const uint32_t*>(es + (off - mis));
        ring = ring_lane;
        pos = mis * 8;
        wr = 0;
        topup();
        const uint32_t i = pos >> 5;
        hi = ring[(i % kRingDwords) * 64];
        lo = ring[((i + 1) % kRingDwords) * 64];
        nx = ring[((i + 2) % kRingDwords) * 64];
        pend = 0;
        crossed = false;
    }
    // The window lives in registers: dwords i, i + 1, i + 2 of the stream (i = pos >> 5), for EVERY element of the slice
    // (round 2 kept it there inside the coefficient loop only; macroblock headers and block starts read the ring twice
    // per element and were half of a P-slice wave's time, tools/dbg/wave_times.py).  So the chain  position -> window
    // -> table -> length -> position  holds ONE LDS round trip, the table.  An element is at most 28 bits, so an advance
    // crosses at most one dword boundary; the dword that then becomes i + 2 is requested at once (`pend`) and put in
    // place by the NEXT advance, where its latency has long been covered by that element's table look-up.  The ring
    // always holds dword i + 2 (top-ups keep eight dwords ahead).
    uint32_t hi, lo, nx, pend;
    bool crossed;
    // the next 32 bits of the stream, MSB first
    __device__ inline uint32_t window() const
    {
        return (uint32_t)((((((uint64_t)hi) << 32) | lo) << (pos & 31)) >> 32);
    }
    __device__ inline void advance(uint32_t n)  // n <= 32
    {
        nx = crossed ? pend : nx;
        const uint32_t np = pos + n;
        const bool c = ((np ^ pos) >> 5) != 0;
        pos = np;
        hi = c ? lo : hi;
        lo = c ? nx : lo;
        crossed = c;
        pend = ring[(((np >> 5) + 2) % kRingDwords) * 64];
    }
};

struct SharedTables {
    ParseTables t;
    uint32_t ring[4][kRingDwords + 1][64];  // [wave][dword][lane]; row kRingDwords is scratch
};

// motion_vector(), player.cpp:891-910.  At most 11 + 6 bits: one window.
__device__ inline int decode_motion(BitReader& br, const uint16_t* tab, int pred, int r_size, bool& ok)
{
    uint32_t win = br.window();
    uint32_t e = tab[win >> 21];
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

// DC size + differential of an intra block, player.cpp:1010-1068 (tables B-5a / B-5b): at most
// 10 + 11 bits, all inside one window.  Returns the DC value (= the new predictor) and the bits used.  (The three
// predictors stay in registers: handed in by reference and picked by block number they became a scratch array.)
__device__ inline int decode_dc(uint32_t win, int blk, int pred, uint32_t& used)
{
    int size, len;
    if (blk < 4) {
        uint32_t pb = win >> 23;
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
    } else {
        uint32_t pb = win >> 22;
        int ones = __clz((int)~(pb << 22));
        if (ones == 0) {
            size = (int)((pb >> 8) & 1);
            len = 2;
        } else {
            size = ones + 1;
            len = size < 10 ? size : 10;
        }
    }
    if (size) {
        int delta = (int)((win << len) >> (32 - size));
        len += size;
        if (delta & (1 << (size - 1)))
            pred += delta;
        else
            pred += (int)((~0u << size) | (uint32_t)(delta + 1));
    }
    used = (uint32_t)len;
    return pred;
}

// One DCT run/level symbol from a 32-bit window and its table entry, BRANCH-FREE (selects only):
// a lone wave issues a dependent instruction only every ~8 cycles, and only straight-line code
// lets the scheduler interleave this off-chain work with the window -> table -> length chain.
struct DctSymbol {
    uint32_t len;  // bits consumed
    uint32_t run;
    int level;     // signed
    bool stop;     // end_of_block ("10", table level 63) or an invalid code (table level 63, 0 bits): the block ends here
};
__device__ inline DctSymbol decode_symbol(uint32_t win, uint32_t ent)
{
    DctSymbol y;
    // the table carries the bits consumed (code + sign; 2 for end_of_block; 0 for an invalid code), so
    // the chain  entry -> length -> position  is one mask and one select long
    const uint32_t len_f = ent & 31, run_f = (ent >> 5) & 31, lev_f = ent >> 10;
    y.stop = lev_f == 63;
    const bool esc = lev_f == 0;  // escape: 6-bit run, 8- or 16-bit level (player.cpp:1092-1099)
    const int lvl_n = ((win << (len_f - 1)) >> 31) ? -(int)lev_f : (int)lev_f;  // sign = last bit of the code
    // the level byte as the int8 it is, or -- first byte 0x00 / 0x80 -- the next byte, minus 256 after 0x80:
    // "ext - 2 * lv8" is that for both (lv8 = 0 or 128)
    const int lv8 = (int)__builtin_amdgcn_ubfe(win, 12, 8), ext = (int)__builtin_amdgcn_ubfe(win, 4, 8);
    const bool two = (lv8 & 0x7F) == 0;
    // (blended with masks: as a select hipcc turns it into a branch inside the symbol loop)
    const int two_mask = (int)((uint32_t)((lv8 & 0x7F) - 1) >> 31) * -1;
    const int lvl_e = ((ext - 2 * lv8) & two_mask) | (__builtin_amdgcn_sbfe((int)win, 12, 8) & ~two_mask);
    y.level = esc ? lvl_e : lvl_n;
    y.run = esc ? __builtin_amdgcn_ubfe(win, 20, 6) : run_f;
    y.len = (esc && two) ? 28u : len_f;
    return y;
}

}  // namespace

#ifdef EFX_DEBUG_WAVES
// development aid (tools/dbg/wave_times.py): per wave of the last k_parse launch, {start, end} of s_memrealtime
// (100 MHz), picture index | type << 8 | hardware id << 16
__device__ unsigned long long g_parse_dbg[8 * 16384];
extern "C" int efx_debug_parse_waves(unsigned long long* dst, size_t n)  // dst == NULL: clear
{
    if (!dst) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_parse_dbg)) != hipSuccess)
            return -1;
        return (int)hipMemset(p, 0, sizeof(g_parse_dbg));
    }
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_parse_dbg), n * sizeof(unsigned long long));
}
#endif

__global__ __launch_bounds__(256) void k_parse(const uint8_t* __restrict__ es, const SliceDesc* __restrict__ descs,
                                               DecodeCounters* __restrict__ counters,
                                               const ParseTables* __restrict__ gtab, MbRec* __restrict__ mbrecs,
                                               uint32_t* __restrict__ coefs, uint32_t* __restrict__ status,
                                               int max_pictures, int epoch, int yield)
{
    // kParseLanes slices per wave: the wave's time is the union of its lanes' control flow (every macroblock, block and
    // symbol trip is paid by all of them), so fewer slices per wave shorten it -- at the price of more waves
    const uint32_t gthread = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t gid = (gthread >> 6) * kParseLanes + (threadIdx.x & 63);
    const bool mine = (threadIdx.x & 63) < kParseLanes && gid < counters->total_slices;
#ifdef EFX_DEBUG_WAVES
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    SliceDesc d = {};
    if (mine)
        d = descs[gid];
    if (!__syncthreads_or(mine))
        return;  // no slice in the whole block: leave before staging tables
    // This kernel is a few thousand long, latency-bound waves and runs next to the (many, short)
    // reconstruction waves of the previous decode call: ask the SIMD arbiter to favour it.
#ifndef EFX_PARSE_PRIO
#define EFX_PARSE_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(EFX_PARSE_PRIO);
    __shared__ SharedTables sh;
    {
        // stage the look-up tables: sizeof(ParseTables) is a multiple of 4
        const uint32_t* src = reinterpret_cast<const uint32_t*>(gtab);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.t);
        for (int i = threadIdx.x; i < (int)(sizeof(ParseTables) / 4); i += blockDim.x)
            dst[i] = src[i];
    }
    const uint32_t pic = d.pic_code_flags & 0xFF;
    __syncthreads();
    if (!mine)
        return;

    const int code = (d.pic_code_flags >> 8) & 0xFF;
    const bool i_picture = ((d.pic_code_flags >> 16) & 3) == 1;
    const int full_pel = (d.pic_code_flags >> 18) & 1;
    const int r_size = (d.pic_code_flags >> 19) & 7;
    const bool custom_q = (d.pic_code_flags >> 22) & 1;  // recorded per macroblock for k_recon's dequantiser

    MbRec* recs = mbrecs + ((size_t)d.stream * max_pictures + pic) * kMbCount;
    const int mb_limit = (int)d.mb_limit;
    uint32_t coef_idx = d.es_off * kCoefsPerEsByte;
    // An entry costs at least 3 bits of slice data, so a slice never outgrows its own region of
    // kCoefsPerEsByte entries per byte; a damaged slice that runs on is parked on its last slot.
    const uint32_t coef_last = (d.es_off + d.es_len) * kCoefsPerEsByte - 1;

    BitReader br;
    br.init(es, d.es_off, &sh.ring[threadIdx.x >> 6][0][threadIdx.x & 63]);

    uint32_t st = 0;
    uint32_t n_coefs = 0, n_mbs = 0;
#ifdef EFX_DEBUG_WAVES
    unsigned long long dbg_loop = 0, dbg_hdr = 0, dbg_c0 = __builtin_readcyclecounter();
#endif

    // slice header, player.cpp:1255-1263
    int mb_addr = (code - 1) * kMbW - 1;  // mb_y = code-2, mb_x = mb_width-1
    int dc_y = 128, dc_cr = 128, dc_cb = 128;
    int mv_h = 0, mv_v = 0;
    int qscale;
    {
        uint32_t win = br.window();
        qscale = (int)(win >> 27);
        br.advance(5);
        while (br.window() >> 31) {  // extra_bit_slice = 1: skip it and 8 bits of information
            br.advance(9);
            br.topup();
        }
        br.advance(1);
    }

    for (int mb = 0;; mb++) {
#ifdef EFX_DEBUG_WAVES
        const unsigned long long dbg_h0 = __builtin_readcyclecounter();
#endif
        br.topup();
        uint32_t win = br.window();
        if ((win >> 9) == 0)  // slice_done(): 23 zero bits, player.cpp:1238-1249
            break;

        // macroblock_address_increment with stuffing (34) and escape (35), player.cpp:1267-1275
        int inc = 0;
        uint32_t e;
        int v;
        bool ok = true;
        for (;;) {
            e = sh.t.mba[win >> 21];
            if (!(e & 15)) {
                ok = false;
                break;
            }
            br.advance(e & 15);
            v = (int)(e >> 4);
            if (v != 34)
                break;
            br.topup();
            win = br.window();
        }
        while (ok && v == 35) {
            inc += 33;
            br.topup();
            e = sh.t.mba[br.window() >> 21];
            if (!(e & 15)) {
                ok = false;
                break;
            }
            br.advance(e & 15);
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
            while (inc > 1 && mb_addr + 1 < mb_limit) {  // skipped macroblocks copy the reference (1283-1288)
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
        if (mb_addr >= mb_limit) {  // past the picture, or into the macroblocks of the next slice (SliceDesc::mb_limit)
            if (mb_limit == kMbCount)
                st |= EFX_STREAM_MB_OVERRUN;
            break;
        }

        // macroblock_type (+ quantiser_scale), player.cpp:1292-1296: at most 6 + 5 bits
        win = br.window();
        int type;
        uint32_t used;
        if (i_picture) {
            if (win >> 31) {
                type = 1;
                used = 1;
            } else if ((win >> 30) == 1) {
                type = 17;
                used = 2;
            } else {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
        } else {
            uint32_t t = sh.t.type_p[win >> 26];
            if (!(t & 7)) {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
            used = t & 7;
            type = (int)(t >> 3);
        }
        const bool intra = type & 1;
        if (type & 0x10) {
            qscale = (int)((win << used) >> 27);
            used += 5;
        }
        br.advance(used);

        const uint32_t mb_coef_base = coef_idx;
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
        // predict(), player.cpp:878-881: full-pel vectors are doubled
        const uint32_t rec_mv = ((uint32_t)(full_pel ? mv_h << 1 : mv_h) & 0xFFFF) | ((uint32_t)(full_pel ? mv_v << 1 : mv_v) << 16);
        const uint32_t rec_flags = (uint32_t)((intra ? 1 : 0) | (qscale << 2) | (custom_q ? 0x80 : 0));

        int cbp = intra ? 63 : 0;
        if (type & 0x02) {
            uint32_t c = sh.t.cbp[br.window() >> 23];
            if (!(c & 15)) {
                st |= EFX_STREAM_BAD_VLC;
                break;
            }
            br.advance(c & 15);
            cbp = (int)(c >> 4);
        }

        // ---- the coded blocks ---------------------------------------------------------------------
        // Per block: [intra: DC] then run/level pairs up to end_of_block (player.cpp:1070-1122).  The
        // symbol decoded in one iteration is stored in the NEXT iteration, in the shadow of that
        // iteration's table look-up (software pipeline of depth one), so the serial chain per symbol
        // is only  window -> table -> length -> position.  The pipeline is seeded with the intra DC
        // entry, or with "1s" = (0, +-1) when a non-intra block opens with a 1 bit (end_of_block
        // cannot come first); when it starts empty the first store lands on a slot that the next
        // real entry overwrites (coef_idx is not advanced).
#ifdef EFX_DEBUG_WAVES
        dbg_hdr += __builtin_readcyclecounter() - dbg_h0;
#endif
        bool bad = false;
        uint32_t cnt_lo = 0, cnt_hi = 0;  // entries per block: blocks 0-3 / 4-5, one byte each
        for (int blk = 0; blk < 6; blk++) {
            if (!(cbp & (0x20 >> blk)))
                continue;
            const uint32_t blk_start = coef_idx;
            br.topup();
            win = br.window();
            uint32_t pend_valid;
            int pend_level, pend_n = 0, n;
            if (intra) {
                uint32_t used;
                pend_level = decode_dc(win, blk, blk < 4 ? dc_y : (blk == 4 ? dc_cr : dc_cb), used);
                dc_y = blk < 4 ? pend_level : dc_y;
                dc_cr = blk == 4 ? pend_level : dc_cr;
                dc_cb = blk == 5 ? pend_level : dc_cb;
                br.advance(used);
                pend_valid = 1;
                n = 1;
            } else {
                pend_valid = win >> 31;
                pend_level = ((win >> 30) & 1) ? -1 : 1;
                n = (int)pend_valid;
                br.advance(pend_valid << 1);
            }
            // The loop leaves nothing behind but its last table entry and position: why a lane stopped -- end_of_block, an
            // invalid code, or a coefficient beyond position 63 (player.cpp:1106-1107: block abandoned) -- is read off
            // them afterwards, so the trip carries one exit test and no flag bookkeeping.
            uint32_t cont, ent;
            int n_new;
#ifdef EFX_DEBUG_WAVES
            const unsigned long long dbg_l0 = __builtin_readcyclecounter();
#endif
            do {
                br.topup();
                win = br.window();
                const uint32_t pk = win >> 16;
                ent = (pk >= 0x0400) ? sh.t.dct_hi[pk >> 8] : sh.t.dct_lo[pk & 0x3FF];
                coefs[min(coef_idx, coef_last)] = ((uint32_t)pend_level << 6) | (uint32_t)pend_n;
                coef_idx += pend_valid;
                const DctSymbol y = decode_symbol(win, ent);
                br.advance(y.len);  // (behind the table look-up: one wait covers both LDS reads)
                n_new = n + (int)y.run;
                cont = !y.stop && n_new < 64;
                pend_valid = cont;
                pend_n = n_new & 63;
                pend_level = y.level;
                n = n_new + 1;
                // When the reconstruction half is the critical path of the pipeline (short slices: the call runs as parse
                // halves with time to spare, efx_decode_range) the parse waves hand issue slots to the k_recon waves they run
                // beside: VALU issue is what the two kernels compete for, and a parse half that finishes
                // early buys nothing.
                if (yield)
                    __builtin_amdgcn_s_sleep(2);
            } while (cont);
#ifdef EFX_DEBUG_WAVES
            dbg_loop += __builtin_readcyclecounter() - dbg_l0;
#endif
            bad = (ent & 31) == 0;                       // invalid code
            const bool dropped = !bad && (ent >> 10) != 63;  // stopped without an end_of_block: ran past position 63
            if (bad)
                break;
            if (dropped) {
                st |= EFX_STREAM_COEF_OVERRUN;
                coef_idx = blk_start;  // forget the partial block
            } else {
                const uint32_t c = coef_idx - blk_start;
                n_coefs += c;
                if (blk < 4)
                    cnt_lo |= c << (8 * blk);
                else
                    cnt_hi |= c << (8 * (blk - 4));
            }
        }
        // MbRec: coef_base | cnt[0..3] | cnt[4] cnt[5] flags epoch | mvx mvy
        *reinterpret_cast<uint4*>(&recs[mb_addr]) =
            make_uint4(mb_coef_base, cnt_lo, cnt_hi | (rec_flags << 16) | ((uint32_t)(epoch & 0xFF) << 24), rec_mv);
        n_mbs++;
        if (bad) {
            st |= EFX_STREAM_BAD_VLC;
            break;
        }
        if (coef_idx > coef_last) {
            st |= EFX_STREAM_BAD_VLC;  // ran past this slice's bytes without finding its end
            break;
        }
    }
#ifdef EFX_DEBUG_WAVES
    {
        // (the last lane of the wave to get here leaves the wave's end time)
        const uint32_t w = gthread >> 6;
        if (w < 16384) {
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            // per wave: start, end, picture | type << 8 | hw id << 16, then the maxima over its lanes of the cycles spent in
            // the symbol loops / in macroblock headers / in ring refills / in all (a lane whose block is not coded sits a
            // symbol loop out: the lane that took part in every one has the wave's figure)
            g_parse_dbg[8 * w] = dbg_t0;
            atomicMax(&g_parse_dbg[8 * w + 1], wall_clock64());
            g_parse_dbg[8 * w + 2] = pic | ((d.pic_code_flags >> 16) & 3) << 8 | ((unsigned long long)hw << 16);
            atomicMax(&g_parse_dbg[8 * w + 3], dbg_loop);
            atomicMax(&g_parse_dbg[8 * w + 4], dbg_hdr);
            atomicMax(&g_parse_dbg[8 * w + 5], br.dbg_topup);
            atomicMax(&g_parse_dbg[8 * w + 6], __builtin_readcyclecounter() - dbg_c0);
        }
    }
#endif
    if (st)
        atomicOr(&status[d.stream], st);
    atomicAdd(&counters->coefficients, (unsigned long long)n_coefs);
    atomicAdd(&counters->macroblocks, (unsigned long long)n_mbs);
}

}  // namespace efx
