// k_tsindex.hip -- trick-play index of a batch of transport streams (gfx950).
//
// Restates the indexer (reference indexer/indexer.cpp): make_index(src, idxs) (86-176) scans the
// 188-byte packets of a stream for video PES headers whose payload starts with a sequence header
// (start code B3) and records (PTS, packet number) for each; pts2seq / pts2pos (180-217) then maps
// every 1/12-second bin of the stream's PTS range to the packet number of the NEAREST sequence
// start -- the random-access table the player seeks with (espflix.cpp:823-829).
//
//   k_ts_sequences  one workgroup per stream, one thread per packet: header parse straight from
//                   global memory, workgroup prefix sum -> ordered (PTS, packet) list, first /
//                   last PTS, and whether the list is sorted (it is, for any real stream);
//   k_idx_bins      one thread per bin: binary search in the sorted list, or the reference's
//                   linear first-minimum scan (with its int truncation) when the list is not
//                   sorted or spans more than 2^31 ticks.
#include <hip/hip_runtime.h>

#include "efx_internal.h"
#include "efx.h"

namespace efx {

namespace {

constexpr int kThreads = 256;

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v)
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

// (int)abs(a - b) of two int64 as the reference computes it (indexer.cpp:188)
__device__ inline int trunc_dist(int64_t a, int64_t b)
{
    int64_t d = a - b;
    return (int)(uint32_t)(uint64_t)(d < 0 ? -d : d);
}

}  // namespace

// seqs: per stream at seqs + pkt_base[s]; entry.es_off = packet number, entry.pts = PTS
__global__ __launch_bounds__(kThreads) void k_ts_sequences(const uint8_t* __restrict__ ts, const uint64_t* __restrict__ stream_off,
                                                           const uint32_t* __restrict__ ts_len,
                                                           const uint32_t* __restrict__ pkt_base, PesEntry* __restrict__ seqs,
                                                           IdxInfo* __restrict__ info)
{
    __shared__ uint32_t sh_wave[4];
    __shared__ int64_t sh_pts[kThreads];   // this chunk's sequence PTS values in order
    __shared__ int64_t sh_last_video;      // PTS of the latest video PES header
    __shared__ int sh_last_tid;
    __shared__ int sh_unsorted;

    const int s = blockIdx.x, tid = threadIdx.x;
    const uint8_t* src = ts + stream_off[s];
    const uint32_t n_packets = ts_len[s] / 188;
    PesEntry* out = seqs + pkt_base[s];
    if (tid == 0) {
        sh_last_video = -1;
        sh_unsorted = 0;
    }
    uint32_t n_seq = 0;
    int64_t origin = -1, prev_pts = 0, min_pts = 0, max_pts = 0;  // prev/min/max valid once n_seq > 0
    __syncthreads();
    for (uint32_t first = 0; first < n_packets; first += kThreads) {
        const uint32_t pkt = first + tid;
        bool vstart = false, is_seq = false;
        int64_t pts = 0;
        if (pkt < n_packets) {
            const uint8_t* d = src + (size_t)pkt * 188;
            const uint32_t pid = ((d[1] << 8) + d[2]) & 0x1FFF;
            int data = 4;
            if (d[3] & 0x20)
                data = 5 + d[4];
            // parse(), indexer.cpp:54-74; a header that leaves the packet is skipped
            if ((d[3] & 0x10) && (d[1] & 0x40) && data + 9 <= 188) {
                const uint8_t* q = d + data + 6;
                const uint32_t flags = (q[0] << 8) | q[1];
                const int payload = data + 9 + q[2];
                q += 3;
                if (flags & 0x0080) {
                    pts = -1;
                    if ((q - d) + 5 <= 188 && (q[0] & 0xF0) == ((flags >> 2) & 0x30)) {
                        pts = ((int64_t)(q[0] & 0x0E)) << 29;
                        pts += (int64_t)((((q[1] << 8) | q[2]) >> 1) << 15);
                        pts += (((q[3] << 8) | q[4]) >> 1);
                    }
                }
                const int marker = payload + 3 < 188 ? d[payload + 3] : -1;
                vstart = pid == 0x100;
                is_seq = vstart && marker == 0xB3;
            }
        }
        // ordered compaction of the sequence starts
        const int wave = tid >> 6, lane = tid & 63;
        const uint32_t incl = wave_incl_scan_u32(is_seq ? 1u : 0u);
        if (lane == 63)
            sh_wave[wave] = incl;
        if (tid == 0)
            sh_last_tid = -1;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave)
                before += sh_wave[w];
            total += sh_wave[w];
        }
        const uint32_t rank = before + incl - (is_seq ? 1u : 0u);
        if (is_seq) {
            PesEntry e;
            e.es_off = pkt;
            e.reserved = 0;
            e.pts = pts;
            out[n_seq + rank] = e;
            sh_pts[rank] = pts;
        }
        if (vstart)
            atomicMax(&sh_last_tid, tid);
        __syncthreads();
        if (vstart && tid == sh_last_tid)
            sh_last_video = pts;  // video_pts = pts for every video PES header, in stream order
        if (is_seq) {
            const int64_t before_pts = rank ? sh_pts[rank - 1] : prev_pts;
            if ((rank || n_seq) && before_pts > pts)
                sh_unsorted = 1;
        }
        if (total) {  // wave-uniform bookkeeping, identical in every thread
            if (!n_seq) {
                origin = sh_pts[0];
                min_pts = max_pts = origin;
            }
            // sorted lists only need the ends; unsorted ones take the slow path anyway
            min_pts = min(min_pts, min(sh_pts[0], sh_pts[total - 1]));
            max_pts = max(max_pts, max(sh_pts[0], sh_pts[total - 1]));
            prev_pts = sh_pts[total - 1];
            n_seq += total;
        }
        __syncthreads();
    }
    if (tid == 0) {
        IdxInfo r;
        r.first_pts = origin;
        r.last_pts = sh_last_video;
        r.n_seq = n_seq;
        // the binary search is exact when the list is sorted and no distance can overflow an int
        const int64_t lo = min(min_pts, min(origin, sh_last_video)), hi = max(max_pts, max(origin, sh_last_video));
        r.fast = n_seq && !sh_unsorted && (hi - lo) < 0x7FFFFFF0ll;
        info[s] = r;
    }
}

// grid = (ceil(samples_cap / 256), streams).  samples[s * samples_cap + b] = pts2pos(first + b * bin)
__global__ __launch_bounds__(kThreads) void k_idx_bins(const PesEntry* __restrict__ seqs, const uint32_t* __restrict__ pkt_base,
                                                       const IdxInfo* __restrict__ info, uint32_t bin_size,
                                                       uint32_t* __restrict__ samples, size_t samples_cap)
{
    const int s = blockIdx.y;
    const IdxInfo r = info[s];
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    const int64_t end = r.last_pts - r.first_pts;
    if (!r.n_seq || end < 0 || (int64_t)b * bin_size > end || b >= samples_cap)
        return;
    const PesEntry* q = seqs + pkt_base[s];
    const int64_t pts = r.first_pts + (int64_t)b * bin_size;
    uint32_t best = 0;
    if (r.fast) {
        // first entry with pts >= target
        uint32_t lo = 0, hi = r.n_seq;
        while (lo < hi) {
            uint32_t m = (lo + hi) >> 1;
            if (q[m].pts < pts)
                lo = m + 1;
            else
                hi = m;
        }
        int64_t dr = lo < r.n_seq ? q[lo].pts - pts : INT64_MAX;
        int64_t dl = lo > 0 ? pts - q[lo - 1].pts : INT64_MAX;
        uint32_t cand = lo;
        int64_t dist = dr;
        if (dl <= dr) {  // the earlier entry wins ties; among equal PTS values the first one
            const int64_t v = q[lo - 1].pts;
            uint32_t a = 0, c = lo - 1;
            while (a < c) {
                uint32_t m = (a + c) >> 1;
                if (q[m].pts < v)
                    a = m + 1;
                else
                    c = m;
            }
            cand = a;
            dist = dl;
        }
        best = dist < 0x7FFFFFF ? cand : 0;  // "mine" starts at 0x7FFFFFF (indexer.cpp:185)
    } else {
        int mine = 0x7FFFFFF;
        for (uint32_t i = 0; i < r.n_seq; i++) {
            int e = trunc_dist(q[i].pts, pts);
            if (e < mine) {
                mine = e;
                best = i;
            }
        }
    }
    samples[(size_t)s * samples_cap + b] = q[best].es_off;
}

}  // namespace efx
