// k_index.hip -- start-code indexing and header parsing (gfx950).
//
// Restates, for a whole batch, what MpegDecoder::run()/marker() do serially per stream
// (reference src/player.cpp:1355-1367,1318-1340): find every 00 00 01 xx start code, read
// sequence (player.cpp:658-678) and picture headers (704-724), and hand each slice
// (0x01..0xAF) to the slice decoder together with the picture state in force.
//
// k_index: one workgroup of kIndexWaves waves per stream.  Lanes read 16 contiguous bytes each (1 KiB per
// wave load, four loads in flight per lane), test 16 byte positions, and append hits to an LDS unit list in
// stream order via wave prefix sums.  Header fields of all units are then pre-parsed lane-parallel
// and the reference's sequential state rules are applied to the whole list at once as counts and
// "most recent unit of a kind" scans.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {

__device__ inline uint32_t wave_excl_scan(uint32_t v, uint32_t* total)
{
    // inclusive Hillis-Steele over 64 lanes
    uint32_t x = v;
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d)
            x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

__device__ inline uint32_t load_bits(const uint8_t* p, uint32_t bitpos, int n)  // n <= 24, MSB first
{
    const uint8_t* q = p + (bitpos >> 3);
    uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    return (w << (bitpos & 7)) >> (32 - n);
}

// The reference does not look for start codes, it HUNTS for markers bit by bit (MpegDecoder::run, player.cpp:1360-1363):
//   while (peek_bits(24) == 0) get_bit();   get_bits(24);   marker(get_bits(8));
// -- the 24 bits are not compared with 00 00 01.  After a header whose fields it has consumed (or ignored: user data,
// extension, a picture header of a type other than I / P, player.cpp:710-717,1328-1330) the hunt therefore arrives at
// the next real start code only if everything in between is zero bits, or is swallowed in 32-bit steps whose last byte
// is a marker value without effect (a slice row beyond the picture, user data, extension, an unknown code).  This
// restates the hunt from stream-relative bit `bit` and says whether it reaches the start code whose value byte is at
// next_off - 1 exactly; if not the reference decodes something else from here on (a phantom picture, a slice parsed
// from the middle of a header ...) which this decoder, indexing byte-aligned start codes, does not reproduce: the
// stream gets EFX_STREAM_SERIAL_HUNT.
__device__ inline bool hunt_arrives(const uint8_t* base, uint32_t bit, uint32_t next_off)
{
    const uint32_t target = next_off * 8;  // position after the next unit's start code
    for (int step = 0; step < 4096; step++) {
        // first set bit at or after `bit` (the '1' of the next start code at the latest)
        uint32_t f = bit;
        {
            uint32_t by = f >> 3;
            uint32_t v = (uint32_t)base[by] & (0xFFu >> (f & 7));
            while (!v && by + 1 < next_off) {
                by++;
                v = base[by];
            }
            if (!v)
                return false;
            f = by * 8 + (uint32_t)(__clz((int)v) - 24);
        }
        const uint32_t at = f - bit >= 24 ? f - 23 : bit;  // zero bits are dropped until the 24-bit window holds a one
        if (at + 32 == target)
            return true;
        if (at + 32 > target)
            return false;
        const uint32_t m = load_bits(base, at + 24, 8);
        if (m < 0x0E || m == 0xB3 || m == 0xB7 || m == 0xB8)
            return false;  // picture / slice / sequence / sequence_end / group: the reference acts on a phantom
        bit = at + 32;
    }
    return false;
}

}  // namespace

__global__ __launch_bounds__(64 * kIndexWaves) void k_index(
    const uint8_t* __restrict__ es, const uint64_t* __restrict__ stream_off, int max_pictures, PicInfo* __restrict__ pics,
    SliceTmp* __restrict__ slices_tmp, uint32_t* __restrict__ pic_count, uint32_t* __restrict__ status, uint32_t* __restrict__ qtab,
    const uint32_t* __restrict__ scan_tab, const PesEntry* __restrict__ pes, const uint32_t* __restrict__ pkt_base,
    const uint32_t* __restrict__ pes_count, int64_t* __restrict__ pts_out, int64_t* __restrict__ pts_newest, int first_picture,
    int stream0, int pic_limit)
{
    __shared__ uint32_t u_off[kMaxUnitsPerStream];
    __shared__ uint32_t u_info[kMaxUnitsPerStream];
    __shared__ uint32_t sh_misc[5];  // pictures; first unit the reference's hunt does not reach; first sequence_end;
                                     // first sequence header of the wrong size; status bits of the walk
    __shared__ uint32_t w_tot[kIndexWaves][4];
    constexpr int kKbPerWave = 16, kKbPerRound = kKbPerWave * kIndexWaves;
    static_assert(kKbPerRound == 64, "one wave scans the kilobyte counts of a round: one per lane");
    __shared__ uint32_t kb_count[kKbPerRound];  // start codes per kilobyte of the round, then their exclusive prefix
    __shared__ uint32_t round_total;

    const int s = stream0 + blockIdx.x;  // (a call runs as groups of streams: efx_decode_from)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* base = es + stream_off[s];
    const uint32_t len = (uint32_t)(stream_off[s + 1] - stream_off[s]);  // padded, multiple of 16

    // ---- 1. start-code scan -------------------------------------------------------------
    // The scan is instructions, not bytes (a wave alone on its SIMD: 117 us for 45 KB when one wave did it all), so
    // the stream is scanned 64 KB at a time by four waves, 16 KB each, a lane 16 bytes of every kilobyte: a 16-bit
    // mask per lane and kilobyte, packed wave scans for the position of a lane's codes inside its kilobyte, the
    // kilobytes' totals through LDS for the position of the kilobyte in the stream's list -- the list is in stream
    // order whatever the wave that found the code.
    uint32_t n_units = 0;
    constexpr int kUnroll = 4;  // four 16-byte loads per lane in flight
    for (uint32_t round = 0; round < len; round += kKbPerRound * 1024) {
        uint32_t mask[kKbPerWave];
        uint32_t before2[kKbPerWave / 2];  // codes before this lane's in its kilobyte, two kilobytes to a register
        const uint32_t wave_base = round + wave * kKbPerWave * 1024;
#pragma unroll
        for (int k = 0; k < kKbPerWave / kUnroll; k++) {
            if (wave_base + k * kUnroll * 1024 >= len) {  // (uniform) nothing of the stream in these four kilobytes
#pragma unroll
                for (int u = 0; u < kUnroll; u++)
                    mask[k * kUnroll + u] = 0;
#pragma unroll
                for (int h = 0; h < kUnroll / 2; h++)
                    before2[(k * kUnroll) / 2 + h] = 0;
                if (lane < kUnroll)
                    kb_count[wave * kKbPerWave + k * kUnroll + lane] = 0;
                continue;
            }
            uint32_t wv[kUnroll][5];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t pos = wave_base + (k * kUnroll + u) * 1024 + lane * 16;
                // reads past the stream end stay inside the ES buffer (next stream / guard) and are masked below
                const uint8_t* p = base + (pos < len ? pos : 0);
                uint4 d = *reinterpret_cast<const uint4*>(p);
                wv[u][0] = d.x;
                wv[u][1] = d.y;
                wv[u][2] = d.z;
                wv[u][3] = d.w;
                wv[u][4] = *reinterpret_cast<const uint32_t*>(p + 16);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t pos = wave_base + (k * kUnroll + u) * 1024 + lane * 16;
                uint32_t* w = wv[u];
                if (pos + 16 >= len)
                    w[4] = 0;  // never look into the next stream
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    // bytes i, i+1, i+2 as a little-endian 24-bit value must be 0x010000
                    uint32_t lo = w[i >> 2], hi = w[(i >> 2) + 1];
                    uint32_t v = (i & 3) ? __builtin_amdgcn_alignbyte(hi, lo, i & 3) : lo;
                    if ((v & 0xFFFFFF) == 0x010000)
                        m |= 1u << i;
                }
                mask[k * kUnroll + u] = pos < len ? m : 0u;
            }
#pragma unroll
            for (int h = 0; h < kUnroll / 2; h++) {
                const int kb = k * kUnroll + 2 * h;
                uint32_t t2;  // (a kilobyte holds at most 342 codes: the two 16-bit fields never carry)
                before2[kb / 2] = wave_excl_scan(__popc(mask[kb]) | (__popc(mask[kb + 1]) << 16), &t2);
                if (lane == 0) {
                    kb_count[wave * kKbPerWave + kb] = t2 & 0xFFFF;
                    kb_count[wave * kKbPerWave + kb + 1] = t2 >> 16;
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t tot;
            const uint32_t ex = wave_excl_scan(kb_count[lane], &tot);
            kb_count[lane] = ex;
            if (lane == 0)
                round_total = tot;
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < kKbPerWave; kb++) {
            uint32_t m = mask[kb];
            // positions in stream order: kilobyte, lane, byte (the start code value is read in phase 2)
            uint32_t idx = n_units + kb_count[wave * kKbPerWave + kb] + ((before2[kb / 2] >> (16 * (kb & 1))) & 0xFFFF);
            const uint32_t pos = wave_base + kb * 1024 + lane * 16;
            while (m) {
                const int i = __ffs(m) - 1;
                m &= m - 1;
                if (idx < kMaxUnitsPerStream)
                    u_off[idx] = pos + i + 4;
                idx++;
            }
        }
        n_units += round_total;
        __syncthreads();  // (kb_count and round_total are written again in the next round)
    }
    uint32_t st = 0;
    if (n_units > kMaxUnitsPerStream) {
        n_units = kMaxUnitsPerStream;
        st |= EFX_STREAM_TOO_MANY_UNITS;
    }
    __syncthreads();

    // ---- 2. lane-parallel header pre-parse -------------------------------------------------
    // sh_misc[1]: index of the first unit from which the reference's marker hunt does not get to the next one
    // (hunt_arrives; 0 also stands for whatever precedes the first start code)
    if (tid == 0) {
        sh_misc[1] = (n_units && !hunt_arrives(base, 0, u_off[0])) ? 0u : ~0u;
        sh_misc[2] = ~0u;
        sh_misc[3] = ~0u;
        sh_misc[4] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_units; i += 64 * kIndexWaves) {
        const uint32_t off = u_off[i];
        const uint8_t* p = base + off;
        const uint32_t code = p[-1];  // the start code value
        u_info[i] = code;  // (stored at once and overwritten below: with the value kept in a register until the end of
                           // the loop body hipcc 7.2 left it undefined for codes B4..B7 -- stale infos, phantom slices)
        // bits of the unit the reference reads before it resumes its marker hunt: user data, extension (player.cpp:1328-1330),
        // unknown codes and slice rows beyond the picture (player.cpp:1255-1258) nothing; a slice in the picture: k_parse's
        // business
        uint32_t consumed = code >= 0x0E ? 0u : ~0u;
        if (code == 0x00) {  // picture: temporal_reference 10, type 3, vbv_delay 16, [full_pel 1, f_code 3]
            uint32_t type = load_bits(p, 10, 3);
            uint32_t fp = load_bits(p, 29, 1), fc = load_bits(p, 30, 3);
            u_info[i] = code | (type << 8) | (fp << 11) | (fc << 12);
            consumed = type == 1 ? 29 : (type == 2 ? 33 : 13);  // player.cpp:704-724
        } else if (code == 0xB3) {  // sequence: 12+12+4+4+18+12 bits, then the two load flags
            uint32_t wdt = load_bits(p, 0, 12), hgt = load_bits(p, 12, 12);
            uint32_t li = load_bits(p, 62, 1);
            uint32_t ln = load_bits(p, li ? 63 + 512 : 63, 1);
            uint32_t bad = (wdt != EFX_FRAME_WIDTH || hgt != EFX_FRAME_HEIGHT);
            u_info[i] = code | (bad << 16) | (li << 17) | (ln << 18);
            consumed = 64 + 512 * (li + ln);  // player.cpp:658-678
            if (bad)
                atomicMin(&sh_misc[3], i);
        } else if (code == 0xB8)
            consumed = 32;  // player.cpp:680-690
        else if (code == 0xB7)
            atomicMin(&sh_misc[2], i);  // sequence_end: the reference pauses here (player.cpp:1324-1327)
        if (consumed != ~0u && code != 0xB7 && i + 1 < n_units) {
            // (a header longer than its unit -- truncated -- has the reference read into the next unit)
            if (off * 8 + consumed > (u_off[i + 1] - 4) * 8 || !hunt_arrives(base, off * 8 + consumed, u_off[i + 1]))
                atomicMin(&sh_misc[1], i);
        }
    }
    __syncthreads();

    // ---- 3. the reference's sequential state rules, all units at once -------------------------------------------
    // MpegDecoder::marker() applied to the unit list in order is a handful of running quantities, each a count or a
    // "most recent unit of a kind" along the list: the picture a unit belongs to (pictures so far), the sequence header
    // and the P picture header in force (most recent B3 / most recent type-2 picture), a slice's rank in its picture
    // (slice rows since the most recent picture), and three places where everything stops or changes -- the first
    // sequence_end (the reference pauses, player.cpp:1324-1327), the picture that exceeds this call's budget, the first
    // sequence header of the wrong size.  A thread takes one unit per round of 256; counts and "most recent" inside a
    // wave are ballots, between waves and rounds a carry through LDS.  Picture `idx` keeps its slices at
    // [idx * kMaxSlicesPerPicture ...) of the stream's temporary list.  A picture's slice count is written by the unit
    // that ends it (the next picture, or whatever stops the walk).  Pictures before `first_picture` (efx_decode_from: a
    // stream longer than max_pictures is decoded in several passes) count for their state and are dropped.
    PicInfo* mypics = pics + (size_t)s * max_pictures;
    SliceTmp* myslices = slices_tmp + (size_t)s * max_pictures * kMaxSlicesPerPicture;
    {
        const uint32_t hunt_lost_at = sh_misc[1], end_at = min(sh_misc[2], n_units), dead_at = sh_misc[3];
        const uint64_t le = ~0ull >> (63 - lane), lt = le >> 1;
        // carries into this round: pictures so far, most recent B3 / P picture / picture (unit index + 1, 0 = none),
        // slice rows since the most recent picture
        uint32_t c_pics = 0, c_b3 = 0, c_p = 0, c_seg = 0;
        uint32_t my_st = 0;
        for (uint32_t round = 0; round <= end_at; round += 64 * kIndexWaves) {
            const uint32_t i = round + tid;
            const bool real = i < end_at;  // (unit end_at -- the sequence_end, or one past the list -- only ends things)
            const uint32_t info = real ? u_info[i] : 0xFFu, code = info & 0xFF;
            const bool is_pic = code == 0x00, is_p = is_pic && ((info >> 8) & 7) == 2, is_b3 = code == 0xB3;
            // slice(): rows beyond the picture are rejected (player.cpp:1255-1258); nothing after a sequence header of
            // the wrong size is decoded
            const bool srow = code >= 0x01 && code <= 0xAF && (int)code - 2 < kMbH && i < dead_at;
            const uint64_t m_pic = __ballot(is_pic), m_p = __ballot(is_p), m_b3 = __ballot(is_b3), m_s = __ballot(srow);
            const int last_pic_w = m_pic ? 63 - __clzll((long long)m_pic) : -1;  // (wave-uniform)
            if (lane == 0) {
                w_tot[wave][0] = (uint32_t)__popcll(m_pic);
                w_tot[wave][1] = m_b3 ? round + wave * 64 + 64 - (uint32_t)__clzll((long long)m_b3) : 0u;
                w_tot[wave][2] = m_p ? round + wave * 64 + 64 - (uint32_t)__clzll((long long)m_p) : 0u;
                w_tot[wave][3] = (uint32_t)__popcll(last_pic_w >= 0 ? m_s & ~(~0ull >> (63 - last_pic_w)) : m_s) |
                                 (last_pic_w >= 0 ? 1u << 31 : 0u);
            }
            __syncthreads();
            uint32_t pics_before = c_pics, b3 = c_b3, pp = c_p, seg = c_seg;  // at the start of this wave
#pragma unroll
            for (int w = 0; w < kIndexWaves; w++) {
                const uint32_t t0 = w_tot[w][0], t1 = w_tot[w][1], t2 = w_tot[w][2], t3 = w_tot[w][3];
                if (w < wave) {
                    pics_before += t0;
                    b3 = t1 ? t1 : b3;
                    pp = t2 ? t2 : pp;
                    seg = (t3 >> 31) ? (t3 & 0xFFFF) : seg + (t3 & 0xFFFF);
                }
                c_pics += t0;
                c_b3 = t1 ? t1 : c_b3;
                c_p = t2 ? t2 : c_p;
                c_seg = (t3 >> 31) ? (t3 & 0xFFFF) : c_seg + (t3 & 0xFFFF);
            }
            __syncthreads();  // (w_tot is written again in the next round)
            // this unit's view: pictures before it, B3 / P picture at or before it, slice rows since the picture it is in
            pics_before += (uint32_t)__popcll(m_pic & lt);
            if (m_b3 & le)
                b3 = round + wave * 64 + 64 - (uint32_t)__clzll((long long)(m_b3 & le));
            if (m_p & le)
                pp = round + wave * 64 + 64 - (uint32_t)__clzll((long long)(m_p & le));
            if (m_pic & lt) {
                const int lp = 63 - __clzll((long long)(m_pic & lt));
                seg = (uint32_t)__popcll(m_s & lt & ~(~0ull >> (63 - lp)));
            } else
                seg += (uint32_t)__popcll(m_s & lt);
            if (i > end_at)
                continue;
            // picture index of the picture open before this unit, and of the one a picture header here opens
            const int open = (int)pics_before - 1 - first_picture, opens = open + 1;
            if (pics_before && open >= pic_limit)
                continue;  // a picture header before this unit exceeded the budget: the walk ended there
            const bool over = is_pic && opens >= pic_limit;  // (efx_decode_range: at most pic_limit pictures per stream in this call)
            if (is_pic || i == end_at) {
                if (pics_before && open >= 0)
                    mypics[open].n_slices = (uint16_t)min(seg, (uint32_t)kMaxSlicesPerPicture);
            }
            if (over || i == end_at) {  // exactly one unit ends the walk
                const uint32_t n = (uint32_t)(opens > 0 ? opens : 0);
                pic_count[s] = n;
                sh_misc[0] = n;
                if (over)
                    my_st |= EFX_STREAM_TRUNCATED;
                // the reference's hunt is lost at a unit the walk looked at (the unit that ends it included)
                if (n_units && hunt_lost_at <= min(i, n_units - 1))
                    my_st |= EFX_STREAM_SERIAL_HUNT;
                if (dead_at < i)
                    my_st |= EFX_STREAM_BAD_SIZE;
            } else if (is_pic) {
                if (opens >= 0) {
                    uint32_t p_full = 0, p_r = 0;  // forward_r_size / full_pel_forward persist across pictures
                    if (pp) {
                        const uint32_t pinfo = u_info[pp - 1], fc = (pinfo >> 12) & 7;
                        p_full = (pinfo >> 11) & 1;
                        p_r = fc ? fc - 1 : 0;  // f_code 0 is forbidden; the reference would shift by -1
                    }
                    const uint32_t seq_flags = b3 ? (u_info[b3 - 1] >> 17) & 3 : 0u;
                    const uint32_t type = (info >> 8) & 7;
                    // (field by field: n_slices belongs to the unit that ends the picture)
                    PicInfo* pi = mypics + opens;
                    pi->first_slice = (uint32_t)opens * kMaxSlicesPerPicture;
                    pi->type = (type == 1) ? 1 : 2;
                    pi->full_pel = (uint8_t)p_full;
                    pi->r_size = (uint8_t)p_r;
                    pi->custom_q = seq_flags ? 1 : 0;
                    pi->reserved = (uint16_t)seq_flags;
                    pi->seq_off = b3 ? u_off[b3 - 1] : 0u;
                    pi->start_off = u_off[i];
                }
            } else if (srow && !pics_before) {
                // a slice in front of the first picture header: the reference parses it all the same -- with the P books and
                // the constructor's state (player.cpp:354-369,1251) -- into its current frame, and usually loses its way in
                // it; here it is dropped.  Not the reference's output any more: flagged like a derailed marker hunt.
                my_st |= EFX_STREAM_SERIAL_HUNT;
            } else if (srow && pics_before && open >= 0) {
                if (seg < (uint32_t)kMaxSlicesPerPicture) {
                    const uint32_t next = (i + 1 < n_units) ? u_off[i + 1] - 4 : len;
                    SliceTmp t;
                    t.off = u_off[i];
                    t.len_code = ((next - u_off[i]) << 8) | code;
                    myslices[open * kMaxSlicesPerPicture + seg] = t;
                } else
                    my_st |= EFX_STREAM_TRUNCATED;
            }
        }
        if (my_st)
            atomicOr(&sh_misc[4], my_st);
    }
    __syncthreads();
    if (tid == 0)
        status[s] = st | sh_misc[4];
    __syncthreads();

    // ---- 4. custom quantiser tables (sequence header loaded matrices, player.cpp:666-673) ----
    // The reference stores a loaded matrix in arrival order and later indexes it with the RASTER
    // position zz (player.cpp:646-651,1113); so entry zz of the table is the zz-th byte sent.
    uint32_t npics = sh_misc[0];

    // ---- 3b. PTS latched at each picture header (flush_picture, player.cpp:692-702) -----------
    // The reference latches _pts when picture() runs; by then its bit reader has pulled in the two
    // bytes after the picture_start_code (FILL_BITS look-ahead, player.cpp:348-352), so the PTS in
    // force is that of the newest PES whose payload starts at or before start_off + 1.
    if (pts_out) {
        const PesEntry* mp = pes + pkt_base[s];
        uint32_t np = pes_count[s];
        if (np & kDemuxFailedFlag) {  // (never expected: the one-pass demultiplexer gave up on this stream, k_demux.hip)
            np = 0;
            if (tid == 0)
                atomicOr(&status[s], EFX_STREAM_INTERNAL);
        }
        // the newest PES PTS of this upload, for k_advance: that kernel runs on the reconstruction stream, possibly
        // after a later upload has recycled this upload's lists -- it must not read them itself
        if (tid == 0)
            pts_newest[s] = np ? mp[np - 1].pts : kNoPts;
        for (uint32_t p = tid; p < npics; p += 64 * kIndexWaves) {
            const uint32_t limit = mypics[p].start_off + 1;
            uint32_t a = 0, b = np;  // first entry with es_off > limit
            while (a < b) {
                uint32_t m = (a + b) >> 1;
                if (mp[m].es_off <= limit)
                    a = m + 1;
                else
                    b = m;
            }
            pts_out[(size_t)s * max_pictures + p] = a ? mp[a - 1].pts : -1;
        }
    }

    for (uint32_t p = wave; p < npics; p += kIndexWaves) {  // (one wave per picture, a lane per table entry)
        PicInfo pi = mypics[p];
        if (!pi.custom_q)
            continue;
        uint32_t flags = pi.reserved;  // bit0 load_intra, bit1 load_non_intra
        const uint8_t* sp = base + pi.seq_off;
        uint32_t intra_bit = 63, non_intra_bit = (flags & 1) ? 64 + 512 : 64;
        uint32_t t = scan_tab[lane];
        uint32_t zz = t & 0xFF;
        uint32_t qi = (flags & 1) ? load_bits(sp, intra_bit + 8 * zz, 8) : ((t >> 16) & 0xFF);
        uint32_t qn = (flags & 2) ? load_bits(sp, non_intra_bit + 8 * zz, 8) : 16;
        qtab[((size_t)s * max_pictures + p) * 64 + lane] = (t & 0xFFFF) | (qi << 16) | (qn << 24);
    }
}

// flush_picture() as a per-stream recurrence (reference src/player.cpp:692-702, constructor 354-361): the
// decoder alternates _current / _reference at a picture header only once a picture carried a PES PTS
// (_last_pts != -1), and its frame index, "a PTS has been seen" and the newest PTS survive from one Buffer
// to the next.  One thread per stream turns the pictures of this efx_decode call into ring positions:
//   picture i is reconstructed into slot (pos0 + swaps(i)) % D from slot (pos0 + swaps(i) - 1) % D,
//   swaps(i) = i + 1 if a PTS was seen before this call, else max(0, i - f), f = first picture with a PTS,
// (call_pos[2 s] = pos0, call_pos[2 s + 1] = f or -1), patches the PTS of pictures that precede this
// upload's first PES with the carried one, and advances the state.  It reads hand-over slot memory only (what k_index
// left there), never the upload's buffers: an upload two batches on may already be overwriting those.  Elementary-stream input has no PES
// layer: every picture counts as carrying a PTS (its index).  Runs on the reconstruction stream, which
// orders the calls.
__global__ void k_advance(StreamState* __restrict__ state, const uint32_t* __restrict__ pic_count, int64_t* __restrict__ pts,
                          const int64_t* __restrict__ pts_newest, int stream0, int n_streams, int max_pictures,
                          int32_t* __restrict__ call_pos)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_streams)
        return;
    const int s = stream0 + t;
    StreamState st = state[s];
    const int n = (int)pic_count[s];
    int f = st.pts_seen ? -1 : n;  // first picture of this call whose header latched a PTS
    if (pts) {
        int64_t* my = pts + (size_t)s * max_pictures;
        for (int i = 0; i < n; i++) {
            if (my[i] == -1 && st.pts_carry != -1)
                my[i] = st.pts_carry;  // the PES that carries this picture's PTS arrived with an earlier upload
            if (f == n && my[i] != -1)
                f = i;
        }
        if (pts_newest[s] != kNoPts)  // (k_index: newest PES PTS of the upload this call decoded)
            st.pts_carry = pts_newest[s];
    } else if (!st.pts_seen)
        f = 0;
    call_pos[2 * s] = (int32_t)st.fb_index;
    call_pos[2 * s + 1] = f;
    if (n > 0) {
        st.fb_index += f < 0 ? (uint32_t)n : (uint32_t)(n - 1 > f ? n - 1 - f : 0);
        if (f < n)
            st.pts_seen = 1;
    }
    state[s] = st;
}

// stream_perm: the order in which the streams contribute to a picture index -- by descending
// bitstream length (host side), so that the 64 slices of a parse wave come from streams of similar
// bit rate and finish at similar times.
// Exclusive prefix sum of slice counts over (picture, stream) pairs in picture-major order, so
// that the slices of one picture index are contiguous: a parse wave then holds 64 slices of the
// same picture type.  Single workgroup.
__global__ __launch_bounds__(1024) void k_slice_scan(const PicInfo* __restrict__ pics,
                                                     const uint32_t* __restrict__ pic_count, int n_streams,
                                                     int max_pictures, const uint32_t* __restrict__ stream_perm,
                                                     uint32_t* __restrict__ slice_base,
                                                     DecodeCounters* __restrict__ counters)
{
    // A thread owns a run of consecutive pairs: its loads (permutation -> picture count -> slice count, three dependent
    // trips to memory) are all in flight together, one pass over the workgroup's 1024 run totals orders the runs.
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    constexpr int kRun = 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = n_streams * max_pictures;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * kRun) {
        uint32_t v[kRun];
#pragma unroll
        for (int r = 0; r < kRun; r++) {
            const int i = base + tid * kRun + r;
            v[r] = 0;
            if (i < n) {
                int p = i / n_streams, s = (int)stream_perm[i - p * n_streams];
                if ((uint32_t)p < pic_count[s])
                    v[r] = pics[(size_t)s * max_pictures + p].n_slices;
            }
        }
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < kRun; r++)
            mine += v[r];
        uint32_t tot;
        uint32_t ex = wave_excl_scan(mine, &tot);
        if (lane == 63)
            wave_tot[wv] = tot;
        __syncthreads();
        uint32_t off = carry;
        for (int k = 0; k < wv; k++)
            off += wave_tot[k];
        uint32_t at = off + ex;
#pragma unroll
        for (int r = 0; r < kRun; r++) {
            const int i = base + tid * kRun + r;
            if (i < n)
                slice_base[i] = at;
            at += v[r];
        }
        __syncthreads();
        if (tid == 1023)
            carry = at;
        __syncthreads();
    }
    if (tid == 0) {
        counters->total_slices = carry;
        counters->streams = (uint32_t)n_streams;
        counters->coefficients = 0;
        counters->macroblocks = 0;
        counters->next_wave = 0;
        slice_base[n] = carry;
    }
}

// one thread per (picture, stream, slice slot)
__global__ __launch_bounds__(256) void k_slice_emit(const PicInfo* __restrict__ pics,
                                                    const SliceTmp* __restrict__ slices_tmp,
                                                    const uint32_t* __restrict__ pic_count,
                                                    const uint64_t* __restrict__ stream_off,
                                                    const uint32_t* __restrict__ slice_base, int n_streams,
                                                    int max_pictures, const uint32_t* __restrict__ stream_perm,
                                                    SliceDesc* __restrict__ descs, uint32_t* __restrict__ status)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int k = t % kMaxSlicesPerPicture, i = t / kMaxSlicesPerPicture;
    if (i >= n_streams * max_pictures)
        return;
    int p = i / n_streams, s = (int)stream_perm[i - p * n_streams];
    if ((uint32_t)p >= pic_count[s])
        return;
    PicInfo pi = pics[(size_t)s * max_pictures + p];
    if (k >= pi.n_slices)
        return;
    SliceTmp st = slices_tmp[(size_t)s * max_pictures * kMaxSlicesPerPicture + pi.first_slice + k];
    SliceDesc d;
    d.es_off = (uint32_t)stream_off[s] + st.off;
    d.es_len = st.len_code >> 8;
    d.stream = (uint32_t)s;
    d.pic_code_flags = (uint32_t)p | ((st.len_code & 0xFF) << 8) | ((uint32_t)pi.type << 16) | ((uint32_t)pi.full_pel << 18) |
                       ((uint32_t)pi.r_size << 19) | ((uint32_t)pi.custom_q << 22);
    // A slice starts at column 0 of its row whatever its first address increment says (inc_mb, player.cpp:823-833,1277),
    // so where every slice of the picture starts is known here.  A (damaged) slice that runs on stops at the nearest
    // start, in raster order, of ANY other slice of the picture, and of several slices with the same start code only the
    // last in the bitstream is parsed: every macroblock has exactly one writer among the parse lanes, whatever the
    // damage.  (The reference, one serial decoder, lets whatever comes later in the bitstream overwrite: the same for a
    // slice that runs on into a later slice's row, not for one that runs on into the row of a slice that came EARLIER
    // in the bitstream -- slices out of raster order AND damaged -- a documented deviation)
    uint32_t limit = kMbCount;
    bool out_of_order = false;
    {
        const uint32_t code = st.len_code & 0xFF;
        const SliceTmp* all = slices_tmp + (size_t)s * max_pictures * kMaxSlicesPerPicture + pi.first_slice;
        for (int j = 0; j < (int)pi.n_slices; j++) {
            const uint32_t other = all[j].len_code & 0xFF;
            if (other > code && (other - 1) * kMbW < limit)
                limit = (other - 1) * kMbW;
            if (other == code && j > k)
                limit = 0;  // superseded by a later slice with the same start code
            out_of_order |= j < k && other >= code;
        }
    }
    if (out_of_order)
        atomicOr(&status[s], EFX_STREAM_SLICE_ORDER);  // (rare: damaged or unusual streams only)
    d.mb_limit = limit;
    d.reserved = 0;
    descs[slice_base[i] + k] = d;
}

}  // namespace efx
