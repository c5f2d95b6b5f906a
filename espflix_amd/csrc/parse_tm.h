// parse_tm.h -- the slice parser of k_parse as a per-lane TOKEN MACHINE, lane-level code that also compiles for the host
// (tests/parse_harness.cpp checks it against the parse trace of the test oracle without a GPU).
//
// Restates MpegDecoder::slice() (reference src/player.cpp:1251-1316), motion_vector(s) (891-920) and the entropy-decode
// half of block() (999-1107) in two passes per slice, both run by the slice's lane:
//
//   1. tm_trip(): ONE TRIP = ONE CODE WORD, whatever it is -- an address increment, a macroblock type, a motion code, a
//      coded block pattern, a DC size, a run/level, an end_of_block, a piece of an escape.  The lane is a state machine:
//      its state names a table, the table entry (64 bits, one LDS read) says how many bits the word has, how many raw
//      bits follow it, what it is worth and which state comes next.  There is no code per kind of token: coefficient
//      entries, DC sizes + differential bits and motion codes + residuals all leave as the same 32-bit stream word
//      (extra bits << 16 | value << 6 | scan position); macroblock_type / quantiser_scale / the address increment are
//      ADDED into one header word (`hacc`) at a shift the entry names (tokens that are none of these add into a bit that
//      is never read); every stream word counts into the byte of a 64-bit counter that the lane's place in the macroblock
//      selects.  What follows a macroblock type, a pattern, the second motion code and an end_of_block comes from the
//      macroblock's PLAN (a bit mask: motion, pattern, blocks 0..5) -- the one piece of per-kind logic, and the only
//      place besides the end of a macroblock (one 16-byte record store) where lanes of a wave part ways.
//      The 64 slices of a wave are at 64 different places of their macroblocks; a parser that walks blocks and
//      macroblocks in lock-step (rounds 1-3) retires a symbol in one lane out of five (P pictures: one in seven).  Here
//      every lane retires a code word on every trip.
//   2. tm_finish(): the chains that run ACROSS the macroblocks of a slice, one macroblock per trip: address increments and
//      skipped macroblocks (inc_mb, predict_zero: 1277-1289), quantiser_scale, motion vector prediction with its
//      wrap-around (891-910), DC prediction (1053-1063, reset rules 1280,1302) -- turns the raw records into the MbRec
//      k_recon reads and the DC tokens into absolute DC values.
#pragma once
#include <cstdint>

#include "efx.h"
#include "efx_internal.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EFX_HD __host__ __device__ __forceinline__
#else
#define EFX_HD inline
#endif

namespace efx {

// ---- tables -----------------------------------------------------------------------------------------------------------
// One array of 64-bit entries; a state word = byte offset of its table << 16 | (32 - peek bits): index = window >> shift.
//
// entry.x (low dword)
//   [4:0]   bits of the code word
//   [5]     forward_r_size more raw bits follow (motion codes other than 0)
//   [15:6]  value: coefficient level (sign-magnitude with the sign bit as raw bit, or two's complement: escapes), DC size;
//           for entries that consult the plan: plan bits to add (<< 22: motion, pattern, blocks 0..5 from bit 31 down)
//   [19:16] raw bits behind the code: the sign (1), quantiser_scale (5), the DC differential (= size), 7 of an escape's 8
//   [25:20] run (added to the scan position); header tokens: the value added into hacc
//   [26]    emit a stream word
//   [27]    the next state comes from the macroblock's plan
//   [28]    macroblock_type follows: in I pictures the state word gets the lane's TYPE_I bit
//   [29]    a coefficient: its scan position must stay below 64
//   [30]    a macroblock_type: a macroblock begins (the slice may hold no more)
//   [31]    (copied into the stream word) level = value + raw bits instead of sign-magnitude
// entry.y (high dword) = the next state word, with [11:6] = shift of THIS token's hacc contribution
//
// A bit pattern that is no code word leads to a DEAD state: a two-entry table whose entries consume nothing, change nothing
// and lead back to it -- a lane that has stopped keeps taking trips with its wave at no cost to its state, and WHY it
// stopped is which dead state it is in.  The 23 zero bits that end a slice (slice_done(), player.cpp:1238-1249) are a
// chain of link entries through the address-increment tables that ends in the dead state kDeadEnd.
constexpr uint32_t kTmR = 1u << 5, kTmEmit = 1u << 26, kTmQuery = 1u << 27, kTmTypeFix = 1u << 28, kTmCoef = 1u << 29,
                   kTmType = 1u << 30, kTmAdd = 1u << 31;
constexpr uint32_t kTmOutMask = 0x8000FFC0u;

// stream word: raw bits << 16 | value << 6 | scan position  (bit 31: additive level)
EFX_HD int tm_level(uint32_t w)  // the signed level of a coefficient word (k_recon, harness)
{
    const int v = (int)(w << 16) >> 22;
    const uint32_t x = (w >> 16) & 0x7FF;
    return (w >> 31) ? v + (int)x : ((x & 1) ? -v : v);
}

// header word of a macroblock (64 bits): macroblock_type [5:0] | quantiser_scale [10:6] | address increment [31:12] |
// horizontal motion code + 16 [37:32], its residual [43:38] | vertical [49:44], [55:50] | macroblock_stuffing seen [62:56] |
// [63] never read.  Both counters that a stream can drive as far as it likes -- ISO 11172-2 allows any number of stuffing
// codes and of address escapes, and the reference simply loops (player.cpp:1267-1275) -- sit where a carry cannot reach a
// live field, and tm_guard() keeps them from wrapping: the stuffing count is held between 32 and 95 once it has passed 63
// ("seen" is all that is ever read), an increment of 2^19 and more (the picture has 264 macroblocks) stops the lane.
constexpr int kHaccInc = 12, kHaccStuff = 56, kHaccMvH = 32, kHaccMvV = 44, kHaccNone = 63;

// table placement, in entries
constexpr int kTbDcY1 = 0, kTbDcC1 = 256, kTbDctF = 512, kTbMvH1 = 768, kTbCbp1 = 1024;  // the plan's targets: 2 KB apart
constexpr int kTbTypeP = 1280, kTbMbaA2 = 1344, kTbMbaB2 = 1408, kTbMvH2 = 1472, kTbMvV2 = 1536, kTbEscR = 1600;
constexpr int kTbCbp2 = 1664 /* 3 x 2 */, kTbDcY2 = 1670, kTbDcC2 = 1672 /* 4 */, kTbEscP = 1676, kTbEscN = 1678;
constexpr int kTbEnd3 = 1680, kTbEnd4 = 1696, kTbEnd5 = 1712;  // 16 each: the rest of a slice's 23 closing zero bits
constexpr int kTbTypeI = kTbTypeP + 512;  // = 1792: the TYPE_I bit of a state word is bit 28 (4096 bytes)
constexpr int kTbDct = 2048, kTbDctLo = 2304, kTbMbaA1 = 3328, kTbMbaB1 = 3584, kTbEscL = 3840, kTbMvV1 = 4096;
constexpr int kTbDead = 4352;  // 8 x 2: the dead states (the HIGHEST tables: "alive" is one comparison)
constexpr int kTmEntries = kTbDead + 16;
constexpr uint32_t tm_word(int base, int peek) { return ((uint32_t)base * 8u) << 16 | (uint32_t)(32 - peek); }
constexpr uint32_t kTmTypeIBit = 1u << 28;
static_assert(((kTbTypeP * 8) & 4096) == 0 && kTbTypeI * 8 == kTbTypeP * 8 + 4096, "TYPE_I = TYPE_P + 4096 bytes");
constexpr uint32_t kWMbaA1 = tm_word(kTbMbaA1, 8), kWPlan0 = tm_word(kTbDcY1, 8), kWDct = tm_word(kTbDct, 8);
constexpr uint32_t kTmEndBits = 23;  // zero bits that end a slice (slice_done(), player.cpp:1238-1249)
// why a lane stopped
enum : uint32_t {
    kDeadEnd = 0,       // the 23 zero bits that end a slice
    kDeadBadBlock = 1,  // no such code, inside a block
    kDeadBadHeader = 2, // ... in macroblock_type, a motion code, coded_block_pattern
    kDeadBadMba = 3,    // ... in macroblock_address_increment
    kDeadRanPast = 4,   // the slice's stream words outgrew its region
    kDeadLimit = 5,     // one macroblock more than the slice can hold
    kDeadIdle = 6,      // never had a slice
};
constexpr uint32_t tm_dead(uint32_t why) { return tm_word(kTbDead + 2 * (int)why, 1); }
constexpr uint32_t kWDeadMin = tm_word(kTbDead, 0) & 0xFFFF0000u;
EFX_HD bool tm_alive(uint32_t st) { return st < kWDeadMin; }
EFX_HD uint32_t tm_why(uint32_t st) { return ((st >> 19) - (uint32_t)kTbDead) >> 1; }
// plan position (motion, pattern, blocks 0..5) -> table, as nibbles indexed by the counter byte the position owns:
// bytes 0..5 = blocks, 6 = motion, 7 = pattern
constexpr uint32_t kPlanInter = 0x43222222u, kPlanIntra = 0x43110000u;

struct alignas(8) TmE {
    uint32_t x, y;
};
struct TmTables {
    TmE e[kTmEntries];
};
void build_tm_tables(TmTables* t);

// raw record of a coded macroblock (pass 1 -> pass 2): hacc (64 bits) | entries of blocks 0-3 | entries of blocks 4, 5, flags
constexpr uint32_t kRawOverrun = 1u << 24, kRawBadBlock = 1u << 25, kRawRanPast = 1u << 26;

// direct sink (host; a slice's words beyond its region are dropped)
struct TmDirectSink {
    uint32_t* coefs;
    uint32_t coef_last;
    EFX_HD void put(uint32_t slot, uint32_t w)
    {
        if (slot <= coef_last)
            coefs[slot] = w;
    }
    EFX_HD void commit(uint32_t, uint32_t) {}
    EFX_HD void rewind(uint32_t, uint32_t) {}
};

struct TmSlice {      // constants of a slice
    uint32_t coef_last;  // last stream slot of the slice's region
    uint32_t type_bit;   // kTmTypeIBit in I pictures
    uint32_t r_size;     // forward_f_code - 1
    uint32_t max_mbs;    // coded macroblocks the slice can hold: mb_limit - its first address
};

struct TmLane {       // pass-1 state of a lane
    uint32_t st;      // state word
    uint32_t tok;     // next stream slot
    uint32_t n;       // scan position of the next coefficient
    uint32_t todo;    // the rest of the macroblock's plan, next item in bit 31
    uint32_t c8;      // 8 x the counter byte of the item in progress (mod 64)
    uint64_t hacc;
    uint64_t cnt;
    uint32_t nmb;
};

EFX_HD uint32_t tm_ubfe(uint32_t v, uint32_t off, uint32_t width)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    off &= 31;
    width &= 31;
    return width ? (v >> off) & ((1u << width) - 1) : 0u;
#endif
}
EFX_HD uint32_t tm_ffbh(uint32_t v)  // leading zeros; 0xFFFFFFFF for 0 (v_ffbh_u32)
{
    return v ? (uint32_t)__builtin_clz(v) : 0xFFFFFFFFu;
}

// Between two groups of trips (a group is at most 8 trips: each adds at most 33 to the increment, 1 to the stuffing
// count): true when the lane's state changed -- the caller looks its entry up again.
EFX_HD bool tm_guard(TmLane& L)
{
    uint32_t hi = (uint32_t)(L.hacc >> 32);
    const uint32_t lo = (uint32_t)L.hacc;
    hi -= (hi & 0x40000000u) >> 1;  // stuffing count 64 ... -> 32 ...
    L.hacc = ((uint64_t)hi << 32) | lo;
    const bool far = (lo >> 31) != 0 && L.st < (((uint32_t)kTbDead * 8u) << 16);
    L.st = far ? ((((uint32_t)(kTbDead + 2 * 5) * 8u) << 16) | 31u) : L.st;  // tm_dead(kDeadLimit)
    return far;
}

EFX_HD void tm_begin(TmLane& L, uint32_t tok_base, bool has_slice)
{
    L.st = has_slice ? kWMbaA1 : tm_dead(kDeadIdle);
    L.tok = tok_base;
    L.n = L.todo = L.nmb = 0;
    L.hacc = L.cnt = 0;
    L.c8 = 40;
}

// One code word.  `win` = the next 32 bits of the slice, `e` = the entry of the lane's state for them (address =
// (L.st >> 16) + ((win >> (L.st & 31)) << 3)).  `next(bits, state word)` is called as soon as the token's length and the
// state that follows it are known -- the caller advances its bit reader and starts the next table look-up there, in the
// shadow of which the rest of the trip runs (stream word, counters, the record of a finished macroblock).
// `sink` takes the stream words: put(slot, word) for every trip (a token that emits none writes the slot the next word
// overwrites), commit(next slot, emitted) after it, rewind(old next slot, new next slot) when an abandoned block gives its
// slots back.  A lane in a dead state takes trips like any other: its entries are all zero.
template <class Next, class Sink, class RawStore>
EFX_HD bool tm_trip(TmLane& L, uint32_t win, TmE e, const TmSlice& sp, Next&& next, Sink&& sink, RawStore&& store_raw)
{
    const uint32_t len = e.x & 31;
    const uint32_t xb = ((e.x >> 16) & 15) + ((e.x >> 5) & 1) * sp.r_size;
    const uint32_t tot = len + xb;
    const uint32_t x = tm_ubfe(win, 32 - tot, xb);
    const uint32_t emit = (e.x >> 26) & 1;
    const uint32_t r6 = (e.x >> 20) & 63;
    const uint32_t npos = L.n + r6;
    const uint32_t word = (x << 16) | (e.x & kTmOutMask) | npos;
    const uint32_t slot = L.tok;
    // a coefficient beyond position 63: the caller follows up with tm_overflow() (rare: damaged streams)
    const bool overflow = (e.x & kTmCoef) && npos > 63;
    L.hacc += (uint64_t)(r6 | (x << 6)) << ((e.y >> 6) & 63);
    L.cnt += (uint64_t)emit << (L.c8 & 63);
    L.tok += emit;
    L.n = npos + emit;
    L.st = e.y | (e.x & sp.type_bit);
    bool mb_end = false, ran_past = false;
    if ((e.x & kTmQuery) && !overflow) {
        // the plan: what a macroblock_type / coded_block_pattern adds to it (the value field holds plan bits in their
        // entries, a level in a coefficient's), then its next item
        const uint32_t plan_add = (e.x & kTmCoef) ? 0u : (e.x >> 6) & 0x3FF;
        L.todo |= plan_add << 22;
        const uint32_t b = tm_ffbh(L.todo), sh = b + 1;
        const bool more = L.todo != 0;
        L.todo = more ? L.todo << (sh & 31) : 0u;
        L.c8 += sh << 3;
        const uint32_t plan = ((uint32_t)L.hacc & 1) ? kPlanIntra : kPlanInter;
        L.st = kWPlan0 + (tm_ubfe(plan, (L.c8 & 63) >> 1, 4) << 27);
        L.n = 0;
        mb_end = !more;
        ran_past = L.tok > sp.coef_last;  // (only looked at when a macroblock ends) ran past this slice's bytes without finding its end
        L.st = mb_end ? (ran_past ? tm_dead(kDeadRanPast) : kWMbaA1) : L.st;
        // one macroblock more than the slice can hold (its address lies beyond the picture or in the next slice's rows):
        // pass 2 decides what that means; nothing of it is kept
        L.st = ((e.x & kTmType) && L.nmb >= sp.max_mbs) ? tm_dead(kDeadLimit) : L.st;
    }
    next(tot, L.st);
    sink.put(slot, word);
    sink.commit(L.tok, emit);
    if (mb_end) {  // the macroblock is complete
        store_raw(L.nmb, (uint32_t)L.hacc, (uint32_t)(L.hacc >> 32), (uint32_t)L.cnt,
                  (uint32_t)(L.cnt >> 32) | (ran_past ? kRawRanPast : 0u));
        L.nmb++;
        L.hacc = L.cnt = 0;
        L.c8 = 40;
    }
    return overflow;
}

// The follow-up of a trip that returned true: the block is abandoned (player.cpp:1106-1107) -- its stream words are
// forgotten, except an intra block's DC token, which pass 2 needs for the predictor (marked by a count of 0xFF) -- and the
// plan moves on to what follows it.  Returns the bits the trip must give back (a token that carried its block's
// end_of_block as well consumed two bits the reference never saw); the caller re-reads its window and looks up L.st.
template <class Sink, class RawStore>
EFX_HD uint32_t tm_overflow(TmLane& L, TmE e, const TmSlice& sp, Sink&& sink, RawStore&& store_raw)
{
    const uint32_t sh = L.c8 & 63;
    const uint32_t c = (uint32_t)(L.cnt >> sh) & 0xFF;
    const uint32_t keep = (uint32_t)L.hacc & 1;  // intra
    const uint32_t tok_was = L.tok;
    L.tok -= c - keep;
    sink.rewind(tok_was, L.tok);
    L.cnt = (L.cnt & ~((uint64_t)0xFF << sh)) | ((uint64_t)(keep ? 0xFFu : 0u) << sh) | ((uint64_t)kRawOverrun << 32);
    L.n = 0;
    const uint32_t b = tm_ffbh(L.todo), shp = b + 1;
    const bool more = L.todo != 0;
    L.todo = more ? L.todo << (shp & 31) : 0u;
    L.c8 += shp << 3;
    const uint32_t plan = ((uint32_t)L.hacc & 1) ? kPlanIntra : kPlanInter;
    L.st = kWPlan0 + (tm_ubfe(plan, (L.c8 & 63) >> 1, 4) << 27);
    if (!more) {
        const bool ran_past = L.tok > sp.coef_last;
        L.st = ran_past ? tm_dead(kDeadRanPast) : kWMbaA1;
        store_raw(L.nmb, (uint32_t)L.hacc, (uint32_t)(L.hacc >> 32), (uint32_t)L.cnt,
                  (uint32_t)(L.cnt >> 32) | (ran_past ? kRawRanPast : 0u));
        L.nmb++;
        L.hacc = L.cnt = 0;
        L.c8 = 40;
    }
    return (e.x & kTmQuery) ? 2u : 0u;
}

// After the loop: the record of the macroblock in progress.  A code that does not exist inside a block leaves the
// macroblock with the blocks before it (player.cpp: block() fails, slice() goes on -- into the same wall); inside a
// macroblock header it leaves the address increment for pass 2 (the skipped macroblocks before it exist).
template <class RawStore>
EFX_HD void tm_end(TmLane& L, const TmSlice& sp, RawStore&& store_raw)
{
    const uint32_t why = tm_why(L.st);
    if (why == kDeadBadBlock) {
        const uint32_t sh = L.c8 & 63;
        L.cnt &= ~((uint64_t)0xFF << sh);
        store_raw(L.nmb, (uint32_t)L.hacc, (uint32_t)(L.hacc >> 32), (uint32_t)L.cnt, (uint32_t)(L.cnt >> 32) | kRawBadBlock);
        L.nmb++;
    } else if (why == kDeadBadHeader && L.nmb < sp.max_mbs)
        store_raw(L.nmb, (uint32_t)L.hacc, (uint32_t)(L.hacc >> 32), 0u, 0u);  // (not counted: the header that could not be read)
}

// ---- pass 2 ------------------------------------------------------------------------------------------------------------
struct TmFix {
    int code;          // slice start code
    int mb_limit;
    uint32_t qscale;   // the slice header's quantiser_scale
    uint32_t full_pel, r_size;
    uint32_t rec_flags;  // 0x80 when the picture uses loaded quantiser matrices
    uint32_t epoch;
    uint32_t coef_last;  // last stream slot of the slice's region: pass 2 never reads a DC token from, or writes a DC value
                         // or an MbRec for, a macroblock whose words do not all lie inside it
};

#if defined(__HIPCC__)
using TmU4 = ::uint4;
#else
struct alignas(16) TmU4 {
    uint32_t x, y, z, w;
};
#endif

EFX_HD int tm_motion(int pred, int mcode, uint32_t residual, int r_size)  // motion_vector(), player.cpp:891-910
{
    int d = mcode;
    if (mcode != 0 && r_size != 0) {
        const int a = mcode < 0 ? -mcode : mcode;
        d = ((a - 1) << r_size) + (int)residual + 1;
        if (mcode < 0)
            d = -d;
    }
    const int scale = 1 << r_size;
    int m = pred + d;
    if (m > (scale << 4) - 1)
        m -= scale << 5;
    else if (m < -(scale << 4))
        m += scale << 5;
    return m;
}

// `raw(k)` reads raw record k of the slice; `recs` = the picture's 264 MbRec; `tok_base` = the slice's first stream slot;
// returns the status bits of the slice.  One macroblock per trip; the raw records come two trips ahead and an intra
// macroblock's DC tokens one trip ahead of their use, so that a trip does not wait for memory.
template <class RawLoad>
EFX_HD uint32_t tm_finish(const TmLane& L, const TmFix& fx, uint32_t tok_base, RawLoad&& raw, uint32_t* coefs, TmU4* recs,
                          uint32_t* n_mbs_out, uint32_t* n_coefs_out)
{
    uint32_t status = 0, n_mbs = 0, n_coefs = 0;
    int mb_addr = (fx.code - 1) * kMbW - 1;
    int dc_y = 128, dc_cr = 128, dc_cb = 128, mv_h = 0, mv_v = 0;
    uint32_t qscale = fx.qscale;
    const uint32_t why = tm_why(L.st);
    const bool partial = why == kDeadBadHeader && L.nmb < (uint32_t)(fx.mb_limit - (fx.code - 1) * kMbW);
    const uint32_t n_rec = L.nmb + (partial ? 1u : 0u);
    const TmU4 none = {0, 0, 0, 0};
    auto count = [](const TmU4& r, int b) { return b < 4 ? (r.z >> (8 * b)) & 0xFF : (r.w >> (8 * (b - 4))) & 0xFF; };
    auto words = [&](const TmU4& r) {  // stream words of the macroblock (an abandoned intra block kept its DC token)
        uint32_t t = 0;
        for (int b = 0; b < 6; b++) {
            const uint32_t c = count(r, b);
            t += c == 0xFF ? 1u : c;
        }
        return t;
    };
    auto load_dc = [&](const TmU4& r, uint32_t first, uint32_t (&w)[6]) {
        uint32_t at = first;
        for (int b = 0; b < 6; b++) {
            const uint32_t c = count(r, b);
            w[b] = ((r.x & 1) && c && at <= fx.coef_last) ? coefs[at] : 0u;  // (a macroblock that ran past its region is dropped below)
            at += c == 0xFF ? 1u : c;
        }
    };
    TmU4 r = n_rec > 0 ? raw(0) : none, r1 = n_rec > 1 ? raw(1) : none;
    uint32_t first = tok_base;
    uint32_t dcw[6], dcw1[6];
    load_dc(r, first, dcw);
    bool stopped = false;
    for (uint32_t k = 0; k < n_rec; k++) {
        const TmU4 r2 = k + 2 < n_rec ? raw(k + 2) : none;
        const uint32_t first1 = first + words(r);
        load_dc(r1, first1, dcw1);
        const uint32_t type = r.x & 63, qf = (r.x >> 6) & 31;
        int inc = (int)(r.x >> kHaccInc);  // (20 bits; tm_guard() stops a lane before bit 31 can carry out)
        if (k == 0)
            mb_addr += 1;  // inc_mb() ignores its argument: the first macroblock sits in column 0 (player.cpp:823-833,1277)
        else {
            if (inc > 1) {
                dc_y = dc_cr = dc_cb = 128;  // reset_predictors(), player.cpp:1280-1281
                mv_h = mv_v = 0;
            }
            while (inc > 1 && mb_addr + 1 < fx.mb_limit) {  // skipped macroblocks copy the reference (1283-1288)
                mb_addr++;
                TmU4 o;
                o.x = first;
                o.y = 0;
                o.z = (2u << 16) | ((fx.epoch & 0xFF) << 24);
                o.w = 0;
                recs[mb_addr] = o;
                n_mbs++;
                inc--;
            }
            mb_addr++;
        }
        if (mb_addr >= fx.mb_limit) {
            // past the picture, or into the macroblocks of the next slice: the lane stops here; the reference goes on -- it
            // may end before the next start code (then the later slice overwrites what it wrote: the same frames) or run
            // through it: flagged either way
            status |= EFX_STREAM_MB_OVERRUN;
            stopped = true;
            break;
        }
        if (k == L.nmb) {  // the header that could not be read
            status |= EFX_STREAM_BAD_VLC;
            stopped = true;
            break;
        }
        if ((r.w & kRawRanPast) || first1 > fx.coef_last + 1) {
            // The macroblock's words outgrew the slice's region (pass 1 dropped those beyond it): a damaged slice that ran on
            // through the start codes behind it.  Nothing of it is kept -- the slots past the region belong to the next slice
            // in the bitstream, possibly another stream's, whose lane runs at the same time: a DC value or a shifted word
            // written there would corrupt a stream that is not damaged.
            status |= EFX_STREAM_BAD_VLC;
            stopped = true;
            break;
        }
        const bool intra = type & 1;
        if (type & 0x10)
            qscale = qf;
        if (intra)
            mv_h = mv_v = 0;  // player.cpp:1300
        else {
            dc_y = dc_cr = dc_cb = 128;  // player.cpp:1302
            if (type & 0x08) {
                mv_h = tm_motion(mv_h, (int)(r.y & 63) - 16, (r.y >> 6) & 63, (int)fx.r_size);
                mv_v = tm_motion(mv_v, (int)((r.y >> 12) & 63) - 16, (r.y >> 18) & 63, (int)fx.r_size);
            } else
                mv_h = mv_v = 0;
        }
        uint32_t w_y = r.z, w_z = r.w & 0xFFFF;
        if (intra) {
            // DC tokens (size << 6 | differential bits << 16) -> absolute values (player.cpp:1010-1068), block by block
            uint32_t at = first;
            for (int b = 0; b < 6; b++) {
                const uint32_t c = count(r, b);
                if (!c)
                    continue;
                const uint32_t tkn = dcw[b];
                const uint32_t size = (tkn >> 6) & 0x3FF, bits = (tkn >> 16) & 0x7FF;
                int pred = b < 4 ? dc_y : (b == 4 ? dc_cr : dc_cb);
                if (size) {
                    if (bits & (1u << (size - 1)))
                        pred += (int)bits;
                    else
                        pred += (int)((~0u << size) | (bits + 1));
                }
                dc_y = b < 4 ? pred : dc_y;
                dc_cr = b == 4 ? pred : dc_cr;
                dc_cb = b == 5 ? pred : dc_cb;
                if (c == 0xFF) {
                    // an abandoned block: its DC moved the predictor, the block itself is not reconstructed -- take its
                    // token out of the macroblock's entry list
                    uint32_t rest = 0;
                    for (int c2 = b + 1; c2 < 6; c2++) {
                        const uint32_t cc = count(r, c2);
                        rest += cc == 0xFF ? 1u : cc;
                    }
                    for (uint32_t j = 0; j < rest; j++)
                        coefs[at + j] = coefs[at + j + 1];
                    if (b < 4)
                        w_y &= ~(0xFFu << (8 * b));
                    else
                        w_z &= ~(0xFFu << (8 * (b - 4)));
                    continue;
                }
                coefs[at] = (uint32_t)pred << 6;
                at += c;
            }
        }
        for (int b = 0; b < 6; b++)
            n_coefs += b < 4 ? (w_y >> (8 * b)) & 0xFF : (w_z >> (8 * (b - 4))) & 0xFF;
        // predict(), player.cpp:878-881: full-pel vectors are doubled
        const uint32_t rec_mv = ((uint32_t)(fx.full_pel ? mv_h << 1 : mv_h) & 0xFFFF) | ((uint32_t)(fx.full_pel ? mv_v << 1 : mv_v) << 16);
        const uint32_t rec_flags = (intra ? 1u : 0u) | (qscale << 2) | fx.rec_flags;
        TmU4 o;
        o.x = first;
        o.y = w_y;
        o.z = w_z | (rec_flags << 16) | ((fx.epoch & 0xFF) << 24);
        o.w = rec_mv;
        recs[mb_addr] = o;
        n_mbs++;
        if (r.w & kRawOverrun)
            status |= EFX_STREAM_COEF_OVERRUN;
        if (r.w & (kRawBadBlock | kRawRanPast)) {
            status |= EFX_STREAM_BAD_VLC;
            stopped = true;
            break;
        }
        r = r1;
        r1 = r2;
        first = first1;
        for (int b = 0; b < 6; b++)
            dcw[b] = dcw1[b];
    }
    if (!stopped) {
        // an address increment that does not exist -- or the 23 zero bits that end a slice (slice_done(),
        // player.cpp:1238-1249) met anywhere but where a macroblock would start (after stuffing)
        if (why == kDeadBadMba || (why == kDeadEnd && (((uint32_t)L.hacc >> kHaccInc) != 0 || ((L.hacc >> kHaccStuff) & 0x7F) != 0)))
            status |= EFX_STREAM_BAD_VLC;
        if (why == kDeadLimit || (why == kDeadBadHeader && !partial))
            status |= EFX_STREAM_MB_OVERRUN;  // a macroblock beyond the last one the slice may hold
    }
    *n_mbs_out = n_mbs;
    *n_coefs_out = n_coefs;
    return status;
}

}  // namespace efx
