// k_demux.hip -- MPEG transport stream -> video elementary stream, on the device (gfx950).
//
// Restates MpegDecoder::more() / demux() / parse_pts() (reference src/player.cpp:294-307,381-436,
// 459-493) for a whole batch: the reference walks 188-byte packets one at a time, keeps PID 0x100,
// skips the adaptation field and -- on payload_unit_start -- the PES header at its fixed offsets,
// latching the PES PTS, and hands the remaining payload bytes to the bit reader.  Here one
// workgroup owns one stream and walks it in chunks of kChunk packets:
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
// After the last packet the workgroup appends the reference's end-of-data tail
// 00 | 00 00 01 B7 | 00 00 01 B7 (player.cpp:456,472) and zero-fills the stream's region.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {

constexpr int kChunk = 128;  // packets per workgroup iteration
constexpr int kTsPacket = 188;
constexpr int kThreads = 256;

__device__ inline uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d)
            v += o;
    }
    return v;
}

}  // namespace

// grid = streams, block = 256.  The TS of stream s is the first ts_len[s] bytes at ts + ts_off[s];
// its output goes to es + out_off[s] (for video: the same region, which is fully written: ES, tail,
// zero fill up to out_off[s + 1]).
//
// AUDIO = true extracts what push_audio() receives instead (MpegDecoder::demux, player.cpp:421-433):
// the payloads of PID 0x101 / 0x102 behind the PES header, but only while the LATEST audio PES header
// carried a PTS (_audio_pts != -1).  That gate is a "last defined value" scan over the packets: a
// PES start sets it (open / closed), every other packet inherits it, across chunk boundaries too.
// No lost-sync byte, no end-of-data tail, no PES list.
template <bool AUDIO>
__device__ __forceinline__ void demux_body(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                           const uint32_t* __restrict__ ts_len, const uint32_t* __restrict__ pkt_base,
                                           uint8_t* __restrict__ es, const uint64_t* __restrict__ out_off,
                                           uint32_t* __restrict__ es_len, PesEntry* __restrict__ pes,
                                           uint32_t* __restrict__ pes_count)
{
    __shared__ uint4 sh_pkt4[kChunk * kTsPacket / 16 + 1];
    __shared__ uint32_t sh_prefix[kChunk + 1];  // exclusive prefix of payload bytes; [npk] = chunk total
    __shared__ int32_t sh_src[kChunk];          // LDS byte offset of the payload (-1: emit a zero byte)
    __shared__ uint32_t sh_wave[2][4];          // per-wave totals: payload bytes, PTS flags
    __shared__ uint32_t sh_gate[2];             // audio: last defined gate value of waves 0 and 1

    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const uint8_t* src = ts + ts_off[s];
    uint8_t* dst = es + out_off[s];
    const uint32_t region = AUDIO ? 0u : (uint32_t)(out_off[s + 1] - out_off[s]);
    const uint32_t n_packets = ts_len[s] / kTsPacket;  // a trailing partial packet is never read (player.cpp:459-468)
    PesEntry* my_pes = AUDIO ? nullptr : pes + pkt_base[s];
    uint32_t gate_carry = 1;  // audio: 1 = closed (_audio_pts == -1 at the start), 2 = open
    const uint8_t* pk = reinterpret_cast<const uint8_t*>(sh_pkt4);

    uint32_t es_pos = 0, n_pes = 0;
    for (uint32_t first = 0; first < n_packets; first += kChunk) {
        const uint32_t npk = min((uint32_t)kChunk, n_packets - first);
        // ---- 1. stage the chunk --------------------------------------------------------------
        {
            const uint4* g = reinterpret_cast<const uint4*>(src + (size_t)first * kTsPacket);  // 128 * 188 = 16 * 1504
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
            if (p[0] != 0x47) {
                n = AUDIO ? 0 : 1;  // "ts lost sync": the VIDEO bit reader is handed one zero byte (player.cpp:465-467)
                from = -1;
            } else {
                const uint32_t pid = ((p[1] << 8) + p[2]) & 0x1FFF;
                int pay = 4;
                if (p[3] & 0x20)
                    pay = 5 + p[4];  // adaptation field
                bool keep = (p[3] & 0x10) != 0;
                if (keep && (p[1] & 0x40)) {  // payload_unit_start: PES header at fixed offsets (player.cpp:396-419)
                    if (pay + 9 > kTsPacket)
                        keep = false;
                    else {
                        const uint8_t* q = p + pay + 6;
                        const uint32_t flags = (q[0] << 8) | q[1];
                        pay += 9 + q[2];
                        q += 3;
                        if ((flags & 0x0080) && (q - p) + 5 <= kTsPacket && (q[0] & 0xF0) == ((flags >> 2) & 0x30)) {
                            pts = ((int64_t)(q[0] & 0x0E)) << 29;  // parse_pts, player.cpp:294-307
                            pts += (int64_t)((((q[1] << 8) | q[2]) >> 1) << 15);
                            pts += (((q[3] << 8) | q[4]) >> 1);
                        }
                    }
                }
                if (keep && (AUDIO ? (pid == 0x101 || pid == 0x102) : pid == 0x100)) {
                    has_pts = !AUDIO && pts != -1;
                    if (AUDIO && (p[1] & 0x40))
                        gate = pts != -1 ? 2 : 1;  // _audio_pts = pts at every audio PES start
                    if (pay < kTsPacket) {
                        n = kTsPacket - pay;
                        from = tid * kTsPacket + pay;
                    }
                }
            }
        }
        const int wave = tid >> 6, lane = tid & 63;
        if (AUDIO) {
            // the gate in force at each packet: the last value defined at or before it
            if (wave < 2) {
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(gate, d, 64);
                    if (lane >= d && gate == 0)
                        gate = o;
                }
                if (lane == 63)
                    sh_gate[wave] = gate;
            }
            __syncthreads();
            if (wave == 1 && gate == 0)
                gate = sh_gate[0];
            if (gate == 0)
                gate = gate_carry;
            if (gate != 2)
                n = 0;  // push_audio() is not called while _audio_pts == -1
            const uint32_t last = sh_gate[1] ? sh_gate[1] : sh_gate[0];
            gate_carry = last ? last : gate_carry;
            __syncthreads();
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
        const uint32_t total = sh_wave[0][0] + sh_wave[0][1];
        const uint32_t total_pes = sh_wave[1][0] + sh_wave[1][1];
        if (wave < 2) {
            const uint32_t excl_n = incl_n - n + (wave ? sh_wave[0][0] : 0);
            const uint32_t excl_f = incl_f - has_pts + (wave ? sh_wave[1][0] : 0);
            sh_prefix[tid] = (uint32_t)tid < npk ? excl_n : total;
            sh_src[tid] = from;
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
        // chunk edges go byte by byte)
        const uint32_t lo = es_pos, hi = es_pos + total;
        const uint32_t* pk32 = reinterpret_cast<const uint32_t*>(sh_pkt4);
        for (uint32_t g = (lo >> 4) + tid; g * 16 < hi; g += kThreads) {
            const uint32_t g0 = g * 16;
            const uint32_t first_o = max(g0, lo), last_o = min(g0 + 16, hi);  // bytes [first_o, last_o) of this group
            // packet holding byte first_o: the last j with prefix[j] <= rel (zero-length packets share a prefix)
            uint32_t rel = first_o - lo;
            int a = 0, b = kChunk;  // invariant: prefix[a] <= rel < prefix[b]
#pragma unroll
            for (int it = 0; it < 7; it++) {
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
        es_pos = hi;
        n_pes += total_pes;
        __syncthreads();
    }

    // ---- end of data: tail + zero fill up to the region end (16-byte aligned); video only ------------
    const uint32_t tail_lo = es_pos;
    for (uint32_t d = (tail_lo >> 2) + tid; !AUDIO && d * 4 < region; d += kThreads) {
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
    if (tid == 0) {
        es_len[s] = es_pos;
        if (!AUDIO)
            pes_count[s] = n_pes;
    }
}

// video: MpegDecoder::more() / demux() for PID 0x100; TS and ES of a stream share one region
__global__ __launch_bounds__(kThreads) void k_demux(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ stream_off,
                                                    const uint32_t* __restrict__ ts_len,
                                                    const uint32_t* __restrict__ pkt_base, uint8_t* __restrict__ es,
                                                    uint32_t* __restrict__ es_len, PesEntry* __restrict__ pes,
                                                    uint32_t* __restrict__ pes_count)
{
    demux_body<false>(ts, stream_off, ts_len, pkt_base, es, stream_off, es_len, pes, pes_count);
}

// audio: the byte stream push_audio() receives (PID 0x101 / 0x102)
__global__ __launch_bounds__(kThreads) void k_demux_audio(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ ts_off,
                                                          const uint32_t* __restrict__ ts_len, uint8_t* __restrict__ out,
                                                          const uint64_t* __restrict__ out_off,
                                                          uint32_t* __restrict__ out_len)
{
    demux_body<true>(ts, ts_off, ts_len, nullptr, out, out_off, out_len, nullptr, nullptr);
}

}  // namespace efx
