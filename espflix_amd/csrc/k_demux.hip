// k_demux.hip -- MPEG transport stream -> video elementary stream, on the device (gfx950).
//
// Restates MpegDecoder::more() / demux() / parse_pts() (reference src/player.cpp:294-307,381-436,
// 459-493) for a whole batch: the reference walks 188-byte packets one at a time, keeps PID 0x100,
// skips the adaptation field and -- on payload_unit_start -- the PES header at its fixed offsets,
// latching the PES PTS, and hands the remaining payload bytes to the bit reader.  Here a stream is cut into chunks of
// kChunk packets and a workgroup owns ONE chunk (the chain from chunk to chunk is a prefix over chunk totals, below):
//
//   1. the chunk (kChunk x 188 bytes, 16-byte aligned in the TS buffer) is staged in LDS with
//      coalesced 16-byte loads;
//   2. one thread per packet parses its header out of LDS -> (payload start, payload bytes, PTS);
//   3. a workgroup prefix sum of the payload sizes gives every packet its ES offset (packet order
//      is stream order) and compacts the PES PTS list;
//   4. the ES bytes of the chunk are produced output-centric: one thread per destination group of 16 bytes,
//      the source packet found by binary search in the LDS prefix array; a group inside one payload is five
//      aligned LDS dwords funnel-shifted into one 16-byte store, the others go byte by byte.
//
// Behind the last payload byte k_demux_offsets appends the reference's end-of-data tail
// 00 | 00 00 01 B7 | 00 00 01 B7 (player.cpp:456,472) and zero-fills the stream's region.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {

constexpr int kChunk = kDemuxChunk;  // packets per workgroup
constexpr int kTsPacket = 188;
constexpr int kThreads = kDemuxThreads;
constexpr int kSearchSteps = kChunk > 128 ? 8 : kChunk > 64 ? 7 : kChunk > 32 ? 6 : kChunk > 16 ? 5 : kChunk > 8 ? 4 : 3;
constexpr bool kTwoWaves = kThreads >= 128;  // (a chunk's packets live in the first two waves, or the one there is)
static_assert((1 << kSearchSteps) >= kChunk && kThreads >= 64 && kThreads >= kChunk && kThreads % 64 == 0, "one thread per packet, a binary search over the chunk");

// Wave-wide inclusive scans as six DPP steps (row_shr 1, 2, 4, 8 inside the rows of sixteen lanes, row_bcast 15 and 31 carry
// the rows' totals on; gfx9) -- a __shfl_up() step is a ds_bpermute round trip.
template <int kCtrl, int kRowMask>
__device__ inline uint32_t demux_dpp(uint32_t identity, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, kCtrl, kRowMask, 0xF, false);
}
__device__ inline uint32_t wave_incl_scan(uint32_t v)
{
    v += demux_dpp<0x111, 0xF>(0, v);
    v += demux_dpp<0x112, 0xF>(0, v);
    v += demux_dpp<0x114, 0xF>(0, v);
    v += demux_dpp<0x118, 0xF>(0, v);
    v += demux_dpp<0x142, 0xA>(0, v);
    v += demux_dpp<0x143, 0xC>(0, v);
    return v;
}
__device__ inline uint32_t wave_incl_max(uint32_t v)
{
    v = max(v, demux_dpp<0x111, 0xF>(0, v));
    v = max(v, demux_dpp<0x112, 0xF>(0, v));
    v = max(v, demux_dpp<0x114, 0xF>(0, v));
    v = max(v, demux_dpp<0x118, 0xF>(0, v));
    v = max(v, demux_dpp<0x142, 0xA>(0, v));
    v = max(v, demux_dpp<0x143, 0xC>(0, v));
    return v;
}
// the last value that is not zero among lanes 0 .. this one (values < 4; 0 if none): a maximum over lane << 2 | value
__device__ inline uint32_t wave_last_defined(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63;
    return wave_incl_max(v ? (lane << 2 | v) : 0u) & 3;
}

}  // namespace

// One chunk of kChunk packets is the unit of work; a stream is as many chunks as it needs and ALL chunks of ALL streams run side
// by side (round 5: a workgroup that walked its stream's chunks one after the other was two dependent 128-packet iterations
// long for the benchmark's streams, 0.11 of the HBM roofline; chunks of 16 packets on one wave each: 0.18 -- kDemuxChunk).  What chains from packet to packet in the reference -- the ES
// position, the PES list position, the audio gate -- is a prefix over chunk totals:
//   k_demux_scan     grid (chunks, streams): one thread per packet reads the few header bytes it needs straight from global
//                    memory -> payload bytes, PES-with-PTS count, last audio gate value of the chunk
//   k_demux_offsets  grid streams, one wave: exclusive prefix over the stream's chunks (in place), stream totals, and the
//                    end-of-data tail + zero fill behind the ES (video)
//   k_demux_gather   grid (chunks, streams): the chunk staged in LDS, parsed again, local prefix + the chunk's base, and the
//                    output-centric 16-byte gather
// Chunk records of stream s start at chunk_slot(pkt_base[s], s): a stream of n packets has at most n / kChunk + 1 chunks.
struct DemuxChunk {
    uint32_t bytes;  // scan: payload bytes of the chunk; after k_demux_offsets: ES position of its first payload byte
    uint32_t n_pes;  // ... PES headers with a PTS; after: position in the stream's PES list
    uint32_t gate;   // audio: last defined gate value of the chunk (0 none, 1 closed, 2 open); after: gate in force at its start
    uint32_t pad;
};

__device__ __forceinline__ uint32_t chunk_slot(uint32_t packets_before, int s) { return packets_before / kChunk + (uint32_t)s; }

namespace {

// one packet (188 bytes at p): payload start inside the packet (-1: the video bit reader gets one zero byte), payload bytes,
// PES PTS or -1, audio gate value defined here (0 = none).  MpegDecoder::demux, player.cpp:381-436.
template <bool AUDIO, class Bytes>
__device__ __forceinline__ void parse_packet(Bytes&& p, int32_t& from, uint32_t& n, int64_t& pts, uint32_t& has_pts, uint32_t& gate)
{
    n = 0;
    from = 0;
    pts = -1;
    has_pts = 0;
    gate = 0;
    if (p(0) != 0x47) {
        n = AUDIO ? 0 : 1;  // "ts lost sync": the VIDEO bit reader is handed one zero byte (player.cpp:465-467)
        from = -1;
        return;
    }
    const uint32_t b1 = p(1), b3 = p(3);
    const uint32_t pid = ((b1 << 8) + p(2)) & 0x1FFF;
    int pay = 4;
    if (b3 & 0x20)
        pay = 5 + (int)p(4);  // adaptation field
    bool keep = (b3 & 0x10) != 0;
    if (keep && (b1 & 0x40)) {  // payload_unit_start: PES header at fixed offsets (player.cpp:396-419)
        if (pay + 9 > kTsPacket)
            keep = false;
        else {
            int q = pay + 6;
            const uint32_t flags = (p(q) << 8) | p(q + 1);
            pay += 9 + (int)p(q + 2);
            q += 3;
            if ((flags & 0x0080) && q + 5 <= kTsPacket && (p(q) & 0xF0) == ((flags >> 2) & 0x30)) {
                pts = ((int64_t)(p(q) & 0x0E)) << 29;  // parse_pts, player.cpp:294-307
                pts += (int64_t)((((p(q + 1) << 8) | p(q + 2)) >> 1) << 15);
                pts += (((p(q + 3) << 8) | p(q + 4)) >> 1);
            }
        }
    }
    if (keep && (AUDIO ? (pid == 0x101 || pid == 0x102) : pid == 0x100)) {
        has_pts = !AUDIO && pts != -1;
        if (AUDIO && (b1 & 0x40))
            gate = pts != -1 ? 2 : 1;  // _audio_pts = pts at every audio PES start
        if (pay < kTsPacket) {
            n = (uint32_t)(kTsPacket - pay);
            from = pay;
        }
    }
}

// the audio gate in force at each packet of a chunk (waves 0 and 1 hold the packets): last defined value at or before it,
// `carry` before the chunk.  Returns the packet's gate; *last = the chunk's last defined value (0: none).
__device__ __forceinline__ uint32_t gate_scan(uint32_t gate, uint32_t carry, uint32_t* sh_gate, uint32_t* last)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave < 2) {
        gate = wave_last_defined(gate);
        if (lane == 63)
            sh_gate[wave] = gate;
    }
    __syncthreads();
    if (wave == 1 && gate == 0)
        gate = sh_gate[0];
    *last = (kTwoWaves && sh_gate[1]) ? sh_gate[1] : sh_gate[0];
    if (gate == 0)
        gate = carry;
    __syncthreads();
    return gate;
}

}  // namespace

template <bool AUDIO>
__device__ __forceinline__ void demux_scan_body(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                                const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                DemuxChunk* __restrict__ chunks)
{
    __shared__ uint32_t sh_wave[2][2];
    __shared__ uint32_t sh_gate[2];
    const int s = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_packets = ts_len[s] / kTsPacket, first = blockIdx.x * kChunk;
    if (first >= n_packets)
        return;
    const uint32_t npk = min((uint32_t)kChunk, n_packets - first);
    uint32_t n = 0, has_pts = 0, gate = 0;
    if ((uint32_t)tid < npk) {
        const uint8_t* g = ts + ts_off[s] + (size_t)(first + tid) * kTsPacket;
        int32_t from;
        int64_t pts;
        parse_packet<AUDIO>([&](int i) { return (uint32_t)g[i]; }, from, n, pts, has_pts, gate);
    }
    uint32_t last = 0;
    if (AUDIO) {
        // a packet's payload counts only while the gate is open; what the chunk knows of it: the gate inside the chunk after a
        // PES start.  Packets BEFORE the chunk's first PES start depend on the chunk's incoming gate -- counted separately.
        const uint32_t g_in_chunk = gate_scan(gate, 0, sh_gate, &last);
        // bytes that count if the incoming gate is open (packets with no gate defined yet) go to `pad`, the rest to `bytes`
        const uint32_t open_now = g_in_chunk == 2 ? n : 0, pending = g_in_chunk == 0 ? n : 0;
        const uint32_t a = wave < 2 ? wave_incl_scan(open_now) : 0, b = wave < 2 ? wave_incl_scan(pending) : 0;
        if (wave < 2 && lane == 63) {
            sh_wave[0][wave] = a;
            sh_wave[1][wave] = b;
        }
        __syncthreads();
        if (tid == 0) {
            DemuxChunk c;
            c.bytes = sh_wave[0][0] + (kTwoWaves ? sh_wave[0][1] : 0u);
            c.n_pes = 0;
            c.gate = last;
            c.pad = sh_wave[1][0] + (kTwoWaves ? sh_wave[1][1] : 0u);
            chunks[chunk_slot(pkt_base[s], s) + blockIdx.x] = c;
        }
        return;
    }
    const uint32_t a = wave < 2 ? wave_incl_scan(n) : 0, b = wave < 2 ? wave_incl_scan(has_pts) : 0;
    if (wave < 2 && lane == 63) {
        sh_wave[0][wave] = a;
        sh_wave[1][wave] = b;
    }
    __syncthreads();
    if (tid == 0) {
        DemuxChunk c;
        c.bytes = sh_wave[0][0] + (kTwoWaves ? sh_wave[0][1] : 0u);
        c.n_pes = sh_wave[1][0] + (kTwoWaves ? sh_wave[1][1] : 0u);
        c.gate = 0;
        c.pad = 0;
        chunks[chunk_slot(pkt_base[s], s) + blockIdx.x] = c;
    }
}

// grid = streams, block = 64: the chain across a stream's chunks, then (video) the end of data
template <bool AUDIO>
__device__ __forceinline__ void demux_offsets_body(const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                   DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ es,
                                                   const uint64_t* __restrict__ out_off, uint32_t* __restrict__ es_len,
                                                   uint32_t* __restrict__ pes_count)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const uint32_t n_packets = ts_len[s] / kTsPacket, n_chunks = (n_packets + kChunk - 1) / kChunk;
    DemuxChunk* my = chunks + chunk_slot(pkt_base[s], s);
    uint32_t es_pos = 0, n_pes = 0, gate = 1;  // audio: closed at the start (_audio_pts == -1)
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
        const bool have = c0 + lane < n_chunks;
        DemuxChunk c = {0, 0, 0, 0};
        if (have)
            c = my[c0 + lane];
        // the gate at the start of each chunk: last defined value of the chunks before it (carried in), a wave scan
        const uint32_t g_upto = wave_last_defined(c.gate);
        const uint32_t g_before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g_upto, 0x138, 0xF, 0xF, false);  // wave_shr 1
        const uint32_t g_in = g_before ? g_before : gate;
        const uint32_t bytes = c.bytes + (AUDIO && g_in == 2 ? c.pad : 0u);
        const uint32_t ib = wave_incl_scan(bytes), ip = wave_incl_scan(c.n_pes);
        if (have) {
            DemuxChunk o;
            o.bytes = es_pos + ib - bytes;
            o.n_pes = n_pes + ip - c.n_pes;
            o.gate = g_in;
            o.pad = 0;
            my[c0 + lane] = o;
        }
        es_pos += (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
        n_pes += (uint32_t)__builtin_amdgcn_readlane((int)ip, 63);
        // the gate after this round: the last defined value among its chunks, else unchanged
        const uint32_t tail_gate = (uint32_t)__builtin_amdgcn_readlane((int)g_upto, 63);
        gate = tail_gate ? tail_gate : gate;
    }
    if (lane == 0) {
        es_len[s] = es_pos;
        if (!AUDIO)
            pes_count[s] = n_pes;
    }
    if (AUDIO)
        return;
    // ---- end of data: tail + zero fill up to the region end (16-byte aligned) ---------------------------------------------
    uint8_t* dst = es + out_off[s];
    const uint32_t region = (uint32_t)(out_off[s + 1] - out_off[s]);
    const uint32_t tail_lo = es_pos;
    for (uint32_t d = (tail_lo >> 2) + lane; d * 4 < region; d += 64) {
        const uint32_t o0 = d * 4;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t o = o0 + k;
            if (o >= tail_lo && o < tail_lo + kEsTailBytes) {
                const uint32_t t = o - tail_lo;  // 00 | 00 00 01 B7 | 00 00 01 B7
                const uint32_t byte = (t == 3 || t == 7) ? 0x01u : ((t == 4 || t == 8) ? 0xB7u : 0u);
                word |= byte << (k * 8);
            }
        }
        if (o0 >= tail_lo)
            *reinterpret_cast<uint32_t*>(dst + o0) = word;
        else
            for (uint32_t o = tail_lo; o < o0 + 4; o++)
                dst[o] = (uint8_t)(word >> ((o & 3) * 8));
    }
}

// ---- one pass, one launch (video; round 6) ------------------------------------------------------------------------------------
// The three launches above read the transport stream twice (the scan touches every line for four header bytes a packet) and
// pay two launch boundaries for a prefix over a few chunk totals.  Fused: a chunk's wave stages its packets ONCE, parses them,
// publishes the chunk's totals, and gets its base -- the totals of the stream's chunks before it -- by DECOUPLED LOOK-BACK
// (Merrill & Garland): one 64-bit descriptor per chunk, {state, PES-with-PTS count, payload bytes}, state EMPTY -> AGGREGATE
// (the chunk's own totals, published right after its local scan) -> INCLUSIVE (with everything before it); a wave reads the
// descriptors of up to 64 predecessors at once, adds aggregates back to the first inclusive one, and never waits for more than
// "that predecessor has parsed its 16 packets".
//   * Which chunk a wave takes is a TICKET (one atomic add per wave on the stream's counter), not blockIdx.x: tickets are
//     handed out in the order waves actually start, so every predecessor a wave waits for is running or done -- no order of
//     dispatch is assumed (HIP promises none).  The wait is bounded all the same (wall clock): a lost descriptor flags the
//     stream (EFX_STREAM_INTERNAL) instead of hanging the device.
//   * Descriptors and tickets are 8- / 4-byte agent-scope atomics on both sides (relaxed: the descriptor IS the data; a valid
//     hand-off form of MI355X_MICROARCH.md, "8-B agent atomics both sides"); the host zeroes them before the launch.
//   * The stream's last chunk knows the stream's totals: it writes es_len / pes_count, the end-of-data tail and the zero fill
//     (k_demux_offsets' second half).  A stream without a whole packet is one empty chunk.
constexpr uint64_t kDescAggregate = 1ull << 62, kDescInclusive = 2ull << 62, kDescFailed = 3ull << 62;
constexpr int kDescPesShift = 36;  // bytes [35:0], PES-with-PTS count [61:36]
__device__ __forceinline__ uint64_t desc_pack(uint64_t state, uint32_t bytes, uint32_t n_pes)
{
    return state | ((uint64_t)n_pes << kDescPesShift) | bytes;
}

// `my` = descriptor of chunk `c` of its stream (descriptors 16 bytes apart: the DemuxChunk slots); returns the totals of chunks
// 0 .. c - 1 and publishes chunk c's inclusive totals.  false: gave up waiting.
__device__ __forceinline__ bool demux_look_back(uint64_t* my, uint32_t c, uint32_t bytes, uint32_t n_pes, uint32_t& base_bytes,
                                                uint32_t& base_pes)
{
    const uint32_t lane = threadIdx.x & 63;
    if (lane == 0)
        __hip_atomic_store(my, desc_pack(c == 0 ? kDescInclusive : kDescAggregate, bytes, n_pes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t sum_b = 0, sum_p = 0;
    int32_t idx = (int32_t)c - 1;  // the nearest predecessor not yet accounted for
    const unsigned long long t0 = wall_clock64();
    while (idx >= 0) {
        const int32_t mine = idx - (int32_t)lane;
        uint64_t d = kDescInclusive;  // (lanes beyond chunk 0: nothing)
        if (mine >= 0)
            d = __hip_atomic_load(my - 2 * (ptrdiff_t)(c - (uint32_t)mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot((d >> 62) == 3))  // a predecessor gave up: so does this chunk (the stream's last chunk reports it)
            break;
        const uint64_t empty = __ballot((d >> 62) == 0);
        if (empty) {
            if (wall_clock64() - t0 > 200000000ull)  // 2 s of the 100 MHz counter
                break;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        const uint64_t incl = __ballot((d >> 62) == 2);  // (lanes beyond chunk 0 count as inclusive with nothing in them)
        // lanes 0 .. first inclusive one (or all 64 / down to chunk 0) contribute
        const uint32_t stop = incl ? (uint32_t)__builtin_ctzll(incl) : 63u;
        const bool take = lane <= stop && mine >= 0;
        uint32_t vb = take ? (uint32_t)(d & 0xFFFFFFFFull) : 0u, vp = take ? (uint32_t)((d >> kDescPesShift) & 0x3FFFFFFu) : 0u;
        vb = wave_incl_scan(vb);
        vp = wave_incl_scan(vp);
        sum_b += (uint32_t)__builtin_amdgcn_readlane((int)vb, 63);
        sum_p += (uint32_t)__builtin_amdgcn_readlane((int)vp, 63);
        if (incl) {
            idx = -1;
            break;
        }
        idx -= 64;
    }
    if (idx >= 0) {  // (left the loop without reaching an inclusive descriptor or chunk 0)
        if (lane == 0)
            __hip_atomic_store(my, kDescFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    base_bytes = (uint32_t)sum_b;
    base_pes = (uint32_t)sum_p;
    if (lane == 0 && c != 0)
        __hip_atomic_store(my, desc_pack(kDescInclusive, base_bytes + bytes, base_pes + n_pes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// grid = (chunks, streams), block = kThreads
template <bool AUDIO, bool FUSED = false>
__device__ __forceinline__ void demux_gather_body(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                                  const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                  const DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ es,
                                                  const uint64_t* __restrict__ out_off, PesEntry* __restrict__ pes,
                                                  uint32_t* __restrict__ tickets = nullptr, uint32_t* __restrict__ es_len = nullptr,
                                                  uint32_t* __restrict__ pes_count = nullptr)
{
    __shared__ uint4 sh_pkt4[kChunk * kTsPacket / 16 + 1];
    __shared__ uint32_t sh_prefix[kChunk + 1];  // exclusive prefix of payload bytes; [npk] = chunk total
    __shared__ int32_t sh_src[kChunk];          // LDS byte offset of the payload (-1: emit a zero byte)
    __shared__ uint32_t sh_wave[2][4];          // per-wave totals: payload bytes, PTS flags
    __shared__ uint32_t sh_gate[2];

    const int s = blockIdx.y;
    const int tid = threadIdx.x;
    const uint32_t n_packets = ts_len[s] / kTsPacket;  // a trailing partial packet is never read (player.cpp:459-468)
    uint32_t chunk = blockIdx.x;
    if (FUSED) {
        static_assert(!FUSED || kThreads == 64, "the fused kernel is one wave per chunk");
        uint32_t t = 0;
        if (tid == 0)
            t = atomicAdd(&tickets[s], 1u);
        chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        const uint32_t n_chunks = max(1u, (n_packets + kChunk - 1) / kChunk);  // (a stream without a packet: one empty chunk)
        if (chunk >= n_chunks)
            return;
    }
    const uint32_t first = chunk * kChunk;
    if (!FUSED && first >= n_packets)
        return;
    const uint8_t* src = ts + ts_off[s];
    uint8_t* dst = es + out_off[s];
    PesEntry* my_pes = AUDIO ? nullptr : pes + pkt_base[s];
    DemuxChunk base = {0, 0, 0, 0};
    if (!FUSED)
        base = chunks[chunk_slot(pkt_base[s], s) + chunk];
    uint32_t es_pos = base.bytes, n_pes = base.n_pes;
    const uint8_t* pk = reinterpret_cast<const uint8_t*>(sh_pkt4);
    const uint32_t npk = first < n_packets ? min((uint32_t)kChunk, n_packets - first) : 0u;
    // ---- 1. stage the chunk --------------------------------------------------------------
    {
        const uint4* g = reinterpret_cast<const uint4*>(src + (size_t)first * kTsPacket);  // (kChunk * 188 is a multiple of 16)
        const uint32_t n16 = (npk * kTsPacket + 15) / 16;  // may read <= 15 bytes past the packets: inside the buffer
        for (uint32_t i = tid; i < n16; i += kThreads)
            sh_pkt4[i] = g[i];
    }
    __syncthreads();
    // ---- 2. one thread per packet --------------------------------------------------------
    uint32_t n = 0, has_pts = 0, gate = 0;
    int32_t from = 0;
    int64_t pts = -1;
    if ((uint32_t)tid < npk) {
        const uint8_t* p = pk + tid * kTsPacket;
        parse_packet<AUDIO>([&](int i) { return (uint32_t)p[i]; }, from, n, pts, has_pts, gate);
        if (from > 0 || (from == 0 && n))
            from += tid * kTsPacket;
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (AUDIO) {
        uint32_t last;
        gate = gate_scan(gate, base.gate, sh_gate, &last);
        if (gate != 2)
            n = 0;  // push_audio() is not called while _audio_pts == -1
    }
    // ---- 3. prefix sums over the chunk (packets live in waves 0 and 1) -------------------
    uint32_t incl_n = 0, incl_f = 0;
    if (wave < 2) {
        incl_n = wave_incl_scan(n);
        incl_f = wave_incl_scan(has_pts);
        if (lane == 63) {
            sh_wave[0][wave] = incl_n;
            sh_wave[1][wave] = incl_f;
        }
    }
    __syncthreads();
    const uint32_t total = sh_wave[0][0] + (kTwoWaves ? sh_wave[0][1] : 0u);
    if (FUSED) {
        // the chunk's totals are known: publish, and collect the chunks before it
        uint64_t* my = reinterpret_cast<uint64_t*>(const_cast<DemuxChunk*>(chunks) + chunk_slot(pkt_base[s], s) + chunk);
        const bool ok = demux_look_back(my, chunk, total, sh_wave[1][0], es_pos, n_pes);
        const uint32_t n_chunks = max(1u, (n_packets + kChunk - 1) / kChunk);
        if (!ok) {
            // never expected (a descriptor that did not arrive within two seconds); the failure runs down the stream's chunks
            // and its last chunk leaves an empty stream whose PES count carries the flag k_index turns into EFX_STREAM_INTERNAL
            if (tid == 0 && chunk == n_chunks - 1) {
                es_len[s] = 0;
                pes_count[s] = kDemuxFailedFlag;
            }
            return;
        }
        if (chunk == n_chunks - 1) {
            // the stream's last chunk: its totals, the end-of-data tail and the zero fill (MpegDecoder::more() at the end of
            // data, player.cpp:456,469-473)
            const uint32_t end = es_pos + total;
            if (tid == 0) {
                es_len[s] = end;
                pes_count[s] = n_pes + sh_wave[1][0];
            }
            const uint32_t region = (uint32_t)(out_off[s + 1] - out_off[s]);
            for (uint32_t d = (end >> 2) + tid; d * 4 < region; d += kThreads) {
                const uint32_t o0 = d * 4;
                uint32_t word = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t o = o0 + k;
                    if (o >= end && o < end + kEsTailBytes) {
                        const uint32_t t = o - end;  // 00 | 00 00 01 B7 | 00 00 01 B7
                        const uint32_t byte = (t == 3 || t == 7) ? 0x01u : ((t == 4 || t == 8) ? 0xB7u : 0u);
                        word |= byte << (k * 8);
                    }
                }
                if (o0 >= end)
                    *reinterpret_cast<uint32_t*>(dst + o0) = word;
                else
                    for (uint32_t o = end; o < o0 + 4; o++)
                        dst[o] = (uint8_t)(word >> ((o & 3) * 8));
            }
        }
    }
    if (wave < 2) {
        const uint32_t excl_n = incl_n - n + (wave ? sh_wave[0][0] : 0);
        const uint32_t excl_f = incl_f - has_pts + (wave ? sh_wave[1][0] : 0);
        if (tid < kChunk) {
            sh_prefix[tid] = (uint32_t)tid < npk ? excl_n : total;
            sh_src[tid] = from;
        }
        if (!AUDIO && has_pts) {
            PesEntry e;
            e.es_off = es_pos + excl_n;
            e.reserved = 0;
            e.pts = pts;
            my_pes[n_pes + excl_f] = e;
        }
        if (tid == 0)
            sh_prefix[kChunk] = total;
    }
    __syncthreads();
    // ---- 4. gather: one destination group of 16 bytes per thread --------------------------
    // (one search per group; a group that lies inside one packet's payload -- eleven in twelve -- is five aligned LDS
    // dwords funnel-shifted into four and ONE 16-byte store; the groups across packet boundaries and the ragged
    // chunk edges -- shared with the neighbouring chunks' workgroups, each writing its own bytes -- go byte by byte)
    const uint32_t lo = es_pos, hi = es_pos + total;
    const uint32_t* pk32 = reinterpret_cast<const uint32_t*>(sh_pkt4);
    for (uint32_t g = (lo >> 4) + tid; g * 16 < hi; g += kThreads) {
        const uint32_t g0 = g * 16;
        const uint32_t first_o = max(g0, lo), last_o = min(g0 + 16, hi);  // bytes [first_o, last_o) of this group
        // packet holding byte first_o: the last j with prefix[j] <= rel (zero-length packets share a prefix)
        uint32_t rel = first_o - lo;
        int a = 0, b = kChunk;  // invariant: prefix[a] <= rel < prefix[b]
#pragma unroll
        for (int it = 0; it < kSearchSteps; it++) {
            int m = (a + b) >> 1;
            if (sh_prefix[m] <= rel)
                a = m;
            else
                b = m;
        }
        uint32_t next = sh_prefix[a + 1];
        const int32_t sp0 = sh_src[a];
        if (last_o - first_o == 16 && sp0 >= 0 && next - rel >= 16) {
            const uint32_t at = (uint32_t)sp0 + (rel - sh_prefix[a]), w = at >> 2, sh = (at & 3) * 8;
            const uint32_t d0 = pk32[w], d1 = pk32[w + 1], d2 = pk32[w + 2], d3 = pk32[w + 3], d4 = pk32[w + 4];
            *reinterpret_cast<uint4*>(dst + g0) = make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                                                            __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh));
            continue;
        }
        for (uint32_t o0 = first_o & ~3u; o0 < last_o; o0 += 4) {
            const uint32_t f_o = max(o0, first_o), l_o = min(o0 + 4, last_o);
            uint32_t word = 0;
            for (uint32_t o = f_o; o < l_o; o++, rel++) {
                while (rel >= next) {
                    a++;
                    next = sh_prefix[a + 1];
                }
                const int32_t sp = sh_src[a];
                const uint32_t byte = sp < 0 ? 0u : pk[sp + (rel - sh_prefix[a])];
                word |= byte << ((o & 3) * 8);
            }
            if (l_o - f_o == 4)
                *reinterpret_cast<uint32_t*>(dst + o0) = word;
            else
                for (uint32_t o = f_o; o < l_o; o++)
                    dst[o] = (uint8_t)(word >> ((o & 3) * 8));
        }
    }
}

// video: MpegDecoder::more() / demux() for PID 0x100; TS and ES of a stream share one region
__global__ __launch_bounds__(kThreads) void k_demux_scan(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ stream_off,
                                                         const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                         DemuxChunk* __restrict__ chunks)
{
    demux_scan_body<false>(ts, stream_off, ts_len, pkt_base, chunks);
}
__global__ __launch_bounds__(64) void k_demux_offsets(const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                      DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ es,
                                                      const uint64_t* __restrict__ stream_off, uint32_t* __restrict__ es_len,
                                                      uint32_t* __restrict__ pes_count)
{
    demux_offsets_body<false>(ts_len, pkt_base, chunks, es, stream_off, es_len, pes_count);
}
__global__ __launch_bounds__(kThreads) void k_demux(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ stream_off,
                                                    const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                    const DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ es,
                                                    PesEntry* __restrict__ pes)
{
    demux_gather_body<false>(ts, stream_off, ts_len, pkt_base, chunks, es, stream_off, pes);
}

// video, one pass: grid (chunks, streams), one wave per chunk; `chunks` (descriptors) and `tickets` zeroed by the host
__global__ __launch_bounds__(kThreads) void k_demux_fused(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ stream_off,
                                                          const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                          DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ es,
                                                          PesEntry* __restrict__ pes, uint32_t* __restrict__ tickets,
                                                          uint32_t* __restrict__ es_len, uint32_t* __restrict__ pes_count)
{
    demux_gather_body<false, kThreads == 64>(ts, stream_off, ts_len, pkt_base, chunks, es, stream_off, pes, tickets, es_len, pes_count);
}

// audio: the byte stream push_audio() receives (PID 0x101 / 0x102)
__global__ __launch_bounds__(kThreads) void k_demux_audio_scan(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                                               const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                               DemuxChunk* __restrict__ chunks)
{
    demux_scan_body<true>(ts, ts_off, ts_len, pkt_base, chunks);
}
__global__ __launch_bounds__(64) void k_demux_audio_offsets(const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                            DemuxChunk* __restrict__ chunks, uint32_t* __restrict__ out_len)
{
    demux_offsets_body<true>(ts_len, pkt_base, chunks, nullptr, nullptr, out_len, nullptr);
}
__global__ __launch_bounds__(kThreads) void k_demux_audio(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                                          const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                                          const DemuxChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                          const uint64_t* __restrict__ out_off)
{
    demux_gather_body<true>(ts, ts_off, ts_len, pkt_base, chunks, out, out_off, nullptr);
}

}  // namespace efx
