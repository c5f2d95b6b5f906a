// k_parse.hip -- slice / macroblock / block VLC parse (gfx950): the token machine of parse_tm.h, ONE LANE PER SLICE.
//
// Restates MpegDecoder::slice() (reference src/player.cpp:1251-1316), motion_vector(s) (891-920) and the entropy-decode
// half of block() (999-1107).  Slices are independent: every predictor is reset at the slice header (1260), so a batch of
// B streams x P pictures x S slices gives B*P*S parallel lanes.
//
// Rounds 1-3 walked the macroblocks and blocks of a wave's 64 slices in lock-step: every macroblock, block and symbol
// trip was paid by all lanes, and a P-picture wave kept 13-16 % of its lanes busy (one lane holds an intra macroblock at
// almost every macroblock step).  Here a lane is a table-driven state machine that retires ONE CODE WORD PER TRIP whatever
// the word is (parse_tm.h): the wave's trip count is its longest lane's token count, not the union of its lanes' control
// flow.  A trip is straight-line code -- window, one 64-bit table entry from LDS, add / shift / store -- plus two
// exec-masked regions: the macroblock's plan (what follows a type, a pattern, an end_of_block) and the 16-byte record at
// the end of a macroblock.  The chains that run across macroblocks (address increments and skipped macroblocks, motion
// vector and DC prediction, quantiser_scale) are resolved afterwards by the same lane, one macroblock per trip
// (tm_finish), into the MbRec + coefficient entries k_recon reads:
//   entry = raw bits << 16 | value << 6 | scan position  (level = tm_level(entry); intra DC: DC value << 6 | 0).
//
// Bit reader: a bit POSITION into a per-lane 64-byte LDS ring, refilled by wave-synchronous top-ups (16-byte loads,
// unconditional ring writes: no vmcnt wait inside the loop); the current three dwords sit in registers,
// so the chain  position -> window -> table -> length -> position  holds ONE LDS round trip.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"
#include "parse_tm.h"
#include "efx_probe.h"

namespace efx {

namespace {

constexpr int kRingDwords = 16;  // per-lane bitstream ring in LDS (64 bytes)
#ifndef EFX_TRIPS_PER_TOPUP
#define EFX_TRIPS_PER_TOPUP 4
#endif
constexpr int kTripsPerTopup = EFX_TRIPS_PER_TOPUP;  // trips between two looks at the ring, the stage and who is alive
constexpr int kRingLow = kTripsPerTopup + 4;  // top up when any lane of the wave has fewer dwords than this ahead (a trip consumes less
                                              // than a dword, the window reads two dwords ahead)
#ifndef EFX_DRAIN_WORDS
#define EFX_DRAIN_WORDS 4  // stream words a lane stores at a time: 4 = one 16-byte store; 8 (experiment, round 6) = a whole 32-byte sector
#endif
constexpr int kDrainWords = EFX_DRAIN_WORDS;
constexpr int kStageWords = (kTripsPerTopup <= 4 && kDrainWords == 4) ? 8 : 16;  // parked stream words per lane (a power of two >= trips + drain - 1)

// Each lane owns a ring of kRingDwords big-endian dwords of ITS slice in LDS, laid out ring[k][lane] (a wave's accesses
// hit 64 different banks).  Global memory is read in WAVE-SYNCHRONOUS top-ups: when any lane runs low every lane refills
// its ring to the brim with independent loads, so the HBM/L2 latency is paid once per ~100 code words.
struct BitReader {
    const uint4* gp;     // 16 bytes holding the first byte of this lane's slice (aligned down); global address space
    uint32_t* ring;      // &ring[0][lane]; dword k lives at ring[(k % kRingDwords) * 64]
    uint32_t pos;        // bit position relative to gp
    uint32_t wr;         // dwords copied global -> ring so far (a multiple of 4)
    // the window: dwords i, i + 1 of the stream (i = pos >> 5) and, on its way from the ring since the last advance, dword
    // i + 2.  A token is at most 28 bits, so an advance crosses at most one dword boundary.
    uint32_t hi, lo, nx;

    __device__ inline void topup()
    {
        if (__any((int)(wr - (pos >> 5)) < kRingLow)) {
            // whole 16-byte groups: one load instruction fetches what four did, and a wave's 64 lanes read 64 different
            // lines whatever the width
            const uint32_t groups = (kRingDwords - (wr - (pos >> 5))) >> 2;  // 0..4 (the lane that ran low: >= 2)
#pragma unroll
            for (uint32_t g = 0; g < (uint32_t)kRingDwords / 4; g++) {
                if (!__any(g < groups))
                    break;
                // UNCONDITIONAL load and ring writes (a lane that needs no more aims the surplus at the scratch rows): no
                // load is left pending on any path
                const bool need = g < groups;
                const uint4 v = gp[(wr >> 2) + (need ? g : 0u)];
                const uint32_t slot = need ? (wr + 4 * g) % kRingDwords : (uint32_t)kRingDwords;
                ring[(slot + 0) * 64] = __builtin_bswap32(v.x);
                ring[(slot + 1) * 64] = __builtin_bswap32(v.y);
                ring[(slot + 2) * 64] = __builtin_bswap32(v.z);
                ring[(slot + 3) * 64] = __builtin_bswap32(v.w);
            }
            wr += 4 * groups;
        }
    }
    __device__ inline void init(const uint8_t* __restrict__ es, uint32_t off, uint32_t* ring_lane)
    {
        uint32_t mis = off & 15;  // the ES buffer itself is 256-byte aligned
        gp = reinterpret_cast<const uint4*>(es + (off - mis));
        ring = ring_lane;
        pos = mis * 8;
        wr = 0;
        topup();
        const uint32_t i = pos >> 5;
        hi = ring[(i % kRingDwords) * 64];
        lo = ring[((i + 1) % kRingDwords) * 64];
        nx = ring[((i + 2) % kRingDwords) * 64];
    }
    // the next 32 bits of the stream, MSB first
    __device__ inline uint32_t window() const
    {
        return (uint32_t)((((((uint64_t)hi) << 32) | lo) << (pos & 31)) >> 32);
    }
    __device__ inline void advance(uint32_t n)  // n <= 32
    {
        const uint32_t np = pos + n;
        const bool c = ((np ^ pos) >> 5) != 0;
        pos = np;
        hi = c ? lo : hi;
        lo = c ? nx : lo;
        nx = ring[(((np >> 5) + 2) % kRingDwords) * 64];  // (used by the NEXT advance: a whole trip away)
    }
    __device__ inline void back(uint32_t n)  // gives the last n bits back (rare: the ring still holds their dwords)
    {
        pos -= n;
        const uint32_t i = pos >> 5;
        hi = ring[(i % kRingDwords) * 64];
        lo = ring[((i + 1) % kRingDwords) * 64];
        nx = ring[((i + 2) % kRingDwords) * 64];
    }
};

// Stream words leave in groups of four, every kTripsPerTopup trips: a wave's 64 lanes write 64 different lines whatever a
// lane stores, and a store instruction costs the wave ~0.1 us however few lanes take part -- one dword per lane per trip
// was what a trip cost, and one 16-byte store per trip by the lanes that had just filled a group still a third of it.  A
// lane parks its words in an eight-dword LDS ring; between two groups of trips every lane that has four or more parked
// stores one aligned group.
struct StageSink {
    uint32_t* stage;  // &stage[0][lane], word k of the slice's stream at stage[(k % kStageWords) * 64]
    uint32_t* coefs;
    uint32_t coef_last;  // last slot of the slice's region (regions start and end on multiples of four slots)
    uint32_t flushed;    // slots below this one are in memory (a multiple of four)
    __device__ inline void put(uint32_t slot, uint32_t w) { stage[(slot & (kStageWords - 1)) * 64] = w; }
    __device__ inline void commit(uint32_t, uint32_t) {}
    // between two groups of at most four trips: at most one group has filled up
    __device__ inline void drain(uint32_t next)
    {
#pragma unroll
        for (int rep = 0; rep < (kTripsPerTopup + 3) / 4; rep++)
        if (kDrainWords == 8) {
            // whole aligned 32-byte sectors: regions start (and an abandoned block rewinds) on multiples of four slots, so a lane
            // whose position is an odd multiple of four stores one 16-byte half first
            const uint32_t* g = stage + (flushed & (kStageWords - 4)) * 64;
            if ((flushed & 4) && next - flushed >= 4) {
                if (flushed + 3 <= coef_last)
                    *reinterpret_cast<uint4*>(coefs + (size_t)flushed) = make_uint4(g[0], g[64], g[128], g[192]);
                flushed += 4;
            } else if (next - flushed >= 8) {
                const uint32_t* h = stage + ((flushed + 4) & (kStageWords - 4)) * 64;
                uint4* dst = reinterpret_cast<uint4*>(coefs + (size_t)flushed);
                if (flushed + 3 <= coef_last)
                    dst[0] = make_uint4(g[0], g[64], g[128], g[192]);
                if (flushed + 7 <= coef_last)
                    dst[1] = make_uint4(h[0], h[64], h[128], h[192]);
                flushed += 8;
            }
        } else
        if (next - flushed >= 4) {
            const uint32_t* g = stage + (flushed & (kStageWords - 4)) * 64;
#if defined(EFX_PARSE_ABL) && EFX_PARSE_ABL == 1
            // (ablation, timing only -- wrong pictures: no stream word ever leaves the LDS ring; the upper bound of what any
            // better store pattern could win, profiles/r6_recon_vmem.md)
            if (flushed + 3 <= coef_last && g[0] == 0xDEADBEEFu && g[64] == 0x12345678u)
#else
            if (flushed + 3 <= coef_last)  // (words beyond the slice's region are dropped: the slice is flagged)
#endif
                *reinterpret_cast<uint4*>(coefs + (size_t)flushed) = make_uint4(g[0], g[64], g[128], g[192]);
            flushed += 4;
        }
    }
    __device__ inline void finish(uint32_t next)
    {
        drain(next);
        for (uint32_t k = flushed; k < next; k++)
            if (k <= coef_last)
                coefs[k] = stage[(k & (kStageWords - 1)) * 64];
    }
    __device__ inline void rewind(uint32_t, uint32_t slot)
    {
        // an abandoned block gave slots back: if they reach below what has left, those words come back from memory
        if (slot < flushed) {
            flushed = slot & ~3u;
            for (uint32_t k = flushed; k < slot; k++)
                stage[(k & (kStageWords - 1)) * 64] = coefs[k];
        }
    }
};

struct SharedTables {
    TmTables t;
    uint32_t ring[kParseWaves][kRingDwords + 4][64];  // [wave][dword][lane]; rows kRingDwords ... are scratch
    uint32_t stage[kParseWaves][kStageWords][64];     // [wave][stream word mod kStageWords][lane]
};

}  // namespace

// One group of kParseLanes slices (`w`), one lane per slice: the two passes.
__device__ __forceinline__ void parse_group(uint32_t w, const uint8_t* __restrict__ es, const SliceDesc* __restrict__ descs,
                                            DecodeCounters* __restrict__ counters, SharedTables& sh, MbRec* __restrict__ mbrecs,
                                            TmU4* __restrict__ raw_recs, uint32_t* __restrict__ coefs, uint32_t* __restrict__ status,
                                            int max_pictures, int epoch)
{
    const uint32_t gid = w * kParseLanes + (threadIdx.x & 63);
    const bool mine = (threadIdx.x & 63) < kParseLanes && gid < counters->total_slices;
    SliceDesc d = {};
    if (mine)
        d = descs[gid];
    const uint32_t pic = d.pic_code_flags & 0xFF;
    EFX_PROBE_CLAIM(1, w);
    EFX_PROBE_STAMP(1);
    const int code = (d.pic_code_flags >> 8) & 0xFF;
    const int first_mb = (code - 1) * kMbW;
    const int mb_limit = (int)d.mb_limit;
    // (a slice superseded by a later one with the same start code has limit 0: nothing of it is parsed)
    const bool has_slice = mine && mb_limit > first_mb;

    TmSlice sp;
    sp.coef_last = (d.es_off + d.es_len) * kCoefsPerEsByte - 1;  // (a slice never outgrows its region: kCoefsPerEsByte)
    sp.type_bit = ((d.pic_code_flags >> 16) & 3) == 1 ? kTmTypeIBit : 0u;
    sp.r_size = (d.pic_code_flags >> 19) & 7;
    sp.max_mbs = has_slice ? (uint32_t)(mb_limit - first_mb) : 0u;
    const size_t rec0 = ((size_t)d.stream * max_pictures + pic) * kMbCount;
    TmU4* const raw = raw_recs + rec0 + (has_slice ? first_mb : 0);

    BitReader br;
    br.init(es, d.es_off, &sh.ring[threadIdx.x >> 6][0][threadIdx.x & 63]);

    // slice header, player.cpp:1255-1263
    TmFix fx;
    {
        fx.qscale = br.window() >> 27;
        br.advance(5);
        while (has_slice && (br.window() >> 31)) {  // extra_bit_slice = 1: skip it and 8 bits of information
            br.advance(9);
            br.topup();
        }
        br.advance(1);
    }
    TmLane L;
    const uint32_t tok_base = d.es_off * kCoefsPerEsByte;
    tm_begin(L, tok_base, has_slice);
    auto store_raw = [&](uint32_t k, uint32_t a, uint32_t b, uint32_t c, uint32_t e) { raw[k] = make_uint4(a, b, c, e); };
    const char* const tab = reinterpret_cast<const char*>(&sh.t);
    StageSink sink{&sh.stage[threadIdx.x >> 6][0][threadIdx.x & 63], coefs, sp.coef_last, d.es_off * kCoefsPerEsByte};
    auto lookup = [&](uint32_t st, uint32_t win) {
        return *reinterpret_cast<const TmE*>(tab + (st >> 16) + ((win >> (st & 31)) << 3));
    };

    // ---- pass 1: one code word per lane per trip -----------------------------------------------------------------------
    // The table entry of the NEXT token is requested as soon as this token's length and successor state are known; the
    // rest of the trip runs in the shadow of that LDS round trip.  kTripsPerTopup trips between two looks at the ring (a
    // trip consumes less than a dword and the ring keeps kRingLow ahead) and at who is still alive: a lane that has stopped
    // sits in a dead state and takes the wave's trips without effect.
    EFX_PROBE_STAMP(2);
    EFX_PROBE_SET(6, pic | ((d.pic_code_flags >> 16) & 3) << 8);
    static_assert(kTripsPerTopup + 3 <= kRingLow && kRingLow <= kRingDwords - 4, "the window reads two dwords ahead; a top-up adds whole groups of four");
#ifdef EFX_PROBE
    unsigned efx_probe_trips = 0;
#endif
    uint32_t win = br.window();
    TmE e = lookup(L.st, win);
    do {
        br.topup();
        sink.drain(L.tok);
        if (__any(tm_guard(L))) {  // (garbage only: an address increment beyond any picture)
            win = br.window();
            e = lookup(L.st, win);
        }
#pragma unroll
        for (int t = 0; t < kTripsPerTopup; t++) {
            const uint32_t win0 = win;
            const TmE e0 = e;
            const bool overflow = tm_trip(
                L, win0, e0, sp,
                [&](uint32_t bits, uint32_t st) {
                    br.advance(bits);
                    win = br.window();
                    e = lookup(st, win);
                },
                sink, store_raw);
            if (__any(overflow)) {  // (damaged streams only)
                if (overflow) {
                    br.back(tm_overflow(L, e0, sp, sink, store_raw));
                    win = br.window();
                    e = lookup(L.st, win);
                }
            }
        }
#ifdef EFX_PROBE
        efx_probe_trips += kTripsPerTopup;
#endif
    } while (__any(tm_alive(L.st)));
    EFX_PROBE_STAMP(3);
    EFX_PROBE_SET(5, efx_probe_trips);
    if (!has_slice)
        return;  // (the lane: its wave goes on to the next group when all lanes are back)
    sink.finish(L.tok);
    tm_end(L, sp, store_raw);

    // ---- pass 2: the chains across the macroblocks of the slice ------------------------------------------------------------
    fx.code = code;
    fx.mb_limit = mb_limit;
    fx.full_pel = (d.pic_code_flags >> 18) & 1;
    fx.r_size = sp.r_size;
    fx.rec_flags = ((d.pic_code_flags >> 22) & 1) ? 0x80u : 0u;  // loaded quantiser matrices: recorded per macroblock for k_recon
    fx.epoch = (uint32_t)epoch;
    fx.coef_last = sp.coef_last;
    uint32_t n_mbs = 0, n_coefs = 0;
    uint32_t st = tm_finish(L, fx, tok_base, [&](uint32_t k) { return raw[k]; }, coefs, reinterpret_cast<TmU4*>(mbrecs + rec0),
                            &n_mbs, &n_coefs);
    // A slice ends inside its unit: the 23 zero bits that close it (slice_done(), player.cpp:1238-1249) reach at most up to the
    // '1' of the next start code.  A lane that consumed more has parsed a start code as slice data -- a damaged slice that
    // stayed syntactically valid; the reference, one serial bit reader, runs on from there and never sees the units it
    // swallows: not its output any more.
    if (br.pos - (d.es_off & 15) * 8 > d.es_len * 8 + kTmEndBits)
        st |= EFX_STREAM_BAD_VLC;
    // And after a slice that ended with its 23 zero bits the reference hunts for the next marker bit by bit (run(),
    // player.cpp:1360-1363: skip zero bits, discard 24, take 8): it arrives at the next start code only if everything the
    // slice left unread in its unit is zero.  A damaged slice that met 23 zeros early leaves junk there, the reference takes
    // a phantom marker out of it: flagged like every other derailed hunt.
    if (tm_why(L.st) == kDeadEnd) {
        const uint8_t* unit = es + d.es_off - (d.es_off & 15);  // (br.pos counts from here)
        const uint32_t end = (d.es_off & 15) + d.es_len;
        uint32_t at = br.pos >> 3;
        bool junk = false;
        if (at < end) {
            junk = (unit[at] & (0xFFu >> (br.pos & 7))) != 0;
            for (at++; at < end && !junk; at++)
                junk = unit[at] != 0;
        }
        if (junk)
            st |= EFX_STREAM_SERIAL_HUNT;
    }
    EFX_PROBE_MAX(4, wall_clock64());
    if (st)
        atomicOr(&status[d.stream], st);
    atomicAdd(&counters->coefficients, (unsigned long long)n_coefs);
    atomicAdd(&counters->macroblocks, (unsigned long long)n_mbs);
}

// The kernel's waves PULL their work: a workgroup stages the tables once, then each of its waves takes groups of
// kParseLanes slices off a counter until none is left.  The grid is what the caller wants resident at a time: every parse
// workgroup holds 60 KB of LDS for as long as it lives, and with one wave per group of slices two parse halves kept
// 36 of the chip's 41 MB of LDS away from the k_recon launches they run beside (7 instead of 12 reconstruction waves per
// CU: efx_probe.h, tools/dbg/probe_waves.py) -- those launches, not the parse halves, are what a decode call waits for.
__global__ __launch_bounds__(64 * kParseWaves) void k_parse(const uint8_t* __restrict__ es, const SliceDesc* __restrict__ descs,
                                                            DecodeCounters* __restrict__ counters,
                                                            const TmTables* __restrict__ gtab, MbRec* __restrict__ mbrecs,
                                                            TmU4* __restrict__ raw_recs, uint32_t* __restrict__ coefs,
                                                            uint32_t* __restrict__ status, int max_pictures, int epoch)
{
    const uint32_t n_groups = (counters->total_slices + kParseLanes - 1) / kParseLanes;
    if (blockIdx.x * kParseWaves >= n_groups)
        return;  // (more workgroups than groups of slices: leave before staging tables)
    // a few thousand long, latency-bound waves that run next to the (many, short) reconstruction waves of the previous
    // decode call: ask the SIMD arbiter to favour them
    __builtin_amdgcn_s_setprio(3);
    __shared__ SharedTables sh;
    {
        const uint4* src = reinterpret_cast<const uint4*>(gtab);
        uint4* dst = reinterpret_cast<uint4*>(&sh.t);
        for (int i = threadIdx.x; i < (int)(sizeof(TmTables) / 16); i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();
    for (;;) {
        uint32_t w = 0;
        if ((threadIdx.x & 63) == 0)
            w = atomicAdd(&counters->next_wave, 1u);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= n_groups)
            break;
        parse_group(w, es, descs, counters, sh, mbrecs, raw_recs, coefs, status, max_pictures, epoch);
    }
}

}  // namespace efx

EFX_PROBE_READER(parse)
