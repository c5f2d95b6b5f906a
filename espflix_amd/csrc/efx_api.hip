// efx_api.hip -- the C-ABI of libefx (include/efx.h): context, uploads, kernel launches.
//
// Host orchestration only; all arithmetic of the hot path lives in the k_*.hip kernels.  There
// is no CPU fallback anywhere: without a gfx950 device efx_create fails with
// EFX_ERR_NO_DEVICE / EFX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "efx.h"
#include "efx_internal.h"

namespace efx {
// kernels (k_demux.hip, k_index.hip, k_parse.hip, k_recon.hip, k_video.hip)
__global__ void k_demux(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, uint8_t*, uint32_t*, PesEntry*,
                        uint32_t*);
__global__ void k_demux_audio(const uint8_t*, const uint64_t*, const uint32_t*, uint8_t*, const uint64_t*, uint32_t*);
__global__ void k_ts_sequences(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, PesEntry*, IdxInfo*);
__global__ void k_idx_bins(const PesEntry*, const uint32_t*, const IdxInfo*, uint32_t, uint32_t*, size_t);
__global__ void k_index(const uint8_t*, const uint64_t*, int, PicInfo*, SliceTmp*, uint32_t*, uint32_t*, uint32_t*,
                        const uint32_t*, const PesEntry*, const uint32_t*, const uint32_t*, int64_t*);
__global__ void k_slice_scan(const PicInfo*, const uint32_t*, int, int, const uint32_t*, uint32_t*, DecodeCounters*);
__global__ void k_slice_emit(const PicInfo*, const SliceTmp*, const uint32_t*, const uint64_t*, const uint32_t*, int, int,
                             const uint32_t*, SliceDesc*);
__global__ void k_parse(const uint8_t*, const SliceDesc*, DecodeCounters*, const ParseTables*, MbRec*, uint32_t*, uint32_t*,
                        int, int);
__global__ void k_recon(const MbRec*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*, int, int, int, int, int, int);
__global__ void k_frame_hash(const uint8_t*, int, uint64_t*);
__global__ void k_fill(uint32_t*, uint32_t, size_t);
__global__ void k_composite(const uint8_t*, const VideoTables*, const VideoLineTemplates*, FieldArgs, uint16_t*);
__global__ void k_pdm(const int16_t*, int, int, int32_t*, uint16_t*);
__global__ void k_sbc(const uint8_t*, size_t, int, int, SbcState*, const SbcTables*, int16_t*, size_t, uint32_t*, uint32_t*, int);
}  // namespace efx

using namespace efx;

constexpr int kParseStreams = 2;  // parse halves in flight at once (latency-bound kernels: two overlap well)
constexpr int kTimingRing = 64;   // efx_decode calls whose stage times efx_get_timing can average
constexpr int kSlots = 3;         // parse -> recon hand-over buffer sets (one being reconstructed + two being parsed)

struct efx_ctx {
    efx_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // capacities
    size_t es_cap = 0;  // bytes of ES buffer (including tails and guard)
    int n_streams = 0;  // streams in the current upload
    size_t es_used = 0;
    bool uploaded = false, decoded = false, results_valid = false;

    // device buffers
    uint8_t* d_es = nullptr;
    uint64_t* d_stream_off = nullptr;
    uint32_t* d_stream_perm = nullptr;  // streams by descending length: the order of slices inside a picture index
    ParseTables* d_tables = nullptr;
    uint8_t* d_frames = nullptr;
    // transport-stream input (allocated on the first EFX_FORMAT_TS upload)
    uint8_t* d_ts = nullptr;
    uint32_t* d_ts_len = nullptr;    // per stream: TS bytes
    uint32_t* d_pkt_base = nullptr;  // per stream: first entry of its PES list (= packets before it)
    uint32_t* d_es_len = nullptr;    // per stream: demuxed ES bytes (without tail)
    uint32_t* d_pes_count = nullptr;
    PesEntry* d_pes = nullptr;
    size_t pes_cap = 0;
    // efx_index_streams keeps its own lists: the ones above belong to the uploaded batch (k_index
    // reads them at every efx_decode); only the transient TS staging buffer d_ts is shared
    IdxInfo* d_idx_info = nullptr;
    uint64_t* d_ts_off = nullptr;
    uint32_t* d_idx_len = nullptr;
    uint32_t* d_idx_base = nullptr;
    PesEntry* d_idx_seq = nullptr;
    bool ts_input = false;
    hipEvent_t ev_demux[2] = {nullptr, nullptr};
    float demux_ms = 0.f;
    size_t ts_bytes = 0;
    // kSlots sets of parse -> recon hand-over buffers: efx_decode() number n parses into slot
    // n % kSlots on parse stream n % kParseStreams while the recon stream is still reconstructing
    // earlier calls from the other slots.  The parse half is bound by the latency of its longest
    // waves (3 waves per SIMD, ~1/3 of the issue slots used), so back-to-back decodes keep two parse
    // halves and one reconstruction half on the GPU at once.  Every slot carries its own
    // index / slice-list scratch.
    struct Slot {
        uint32_t* d_pic_count = nullptr;
        uint32_t* d_status = nullptr;
        DecodeCounters* d_counters = nullptr;
        MbRec* d_mbrecs = nullptr;
        uint32_t* d_coefs = nullptr;
        // index / slice-list scratch of this call (per slot: two parse halves run concurrently)
        PicInfo* d_pics = nullptr;
        SliceTmp* d_slices_tmp = nullptr;
        uint32_t* d_qtab = nullptr;  // per (stream, picture) custom quantiser tables, read by k_recon
        uint32_t* d_slice_base = nullptr;
        SliceDesc* d_descs = nullptr;
        int64_t* d_pts = nullptr;  // per (stream, picture): PTS latched at the picture header (TS input)
        hipEvent_t parse_done = nullptr, recon_done = nullptr;
        int epoch = 0;
    } slot[kSlots];
    // stage timing: one event set per efx_decode call since efx_set_timing(1), so that a run of
    // back-to-back (overlapping) calls can be averaged afterwards without a host sync in between
    struct TimingEvents {
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // parse start, index end, parse end, recon end, recon start
    };
    std::vector<TimingEvents> timing_ring;  // kTimingRing sets, created by efx_set_timing
    uint64_t timed_calls = 0;               // calls recorded since timing was (re-)enabled
    int cur = 0;        // slot of the most recent efx_decode
    uint64_t calls = 0;
    hipStream_t parse_streams[kParseStreams] = {nullptr, nullptr};
    VideoTables* d_video[2] = {nullptr, nullptr};  // [0] PAL, [1] NTSC
    VideoLineTemplates* d_video_lines[2] = {nullptr, nullptr};
    SbcTables* d_sbc_tables = nullptr;
    uint64_t* d_hash = nullptr;

    // host staging / results
    uint8_t* h_es = nullptr;  // pinned
    std::vector<uint64_t> h_stream_off;
    std::vector<int64_t> h_pts;  // per (stream, picture), TS input only
    std::vector<uint32_t> h_es_len;  // per stream, ES input only
    std::vector<uint32_t> h_pic_count, h_status;
    DecodeCounters h_counters{};

    bool timing = false;
};

namespace {

int fail(efx_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define EFX_HIP(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess)                                      \
            return fail(ctx, EFX_ERR_DEVICE, #call, e_);           \
    } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t n)
{
    return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}

}  // namespace

extern "C" {

const char* efx_status_string(int status)
{
    switch (status) {
    case EFX_OK: return "ok";
    case EFX_ERR_ARG: return "invalid argument";
    case EFX_ERR_DEVICE: return "HIP runtime error";
    case EFX_ERR_NO_DEVICE: return "no usable gfx950 device";
    case EFX_ERR_CAPACITY: return "context capacity exceeded";
    case EFX_ERR_STATE: return "call out of order";
    case EFX_ERR_STREAM: return "stream violates a decoder constraint";
    default: return "unknown status";
    }
}

const char* efx_last_error(const efx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int efx_create(const efx_config* cfg, efx_ctx** out)
{
    if (!cfg || !out || cfg->max_streams <= 0 || cfg->max_pictures <= 0 || cfg->max_pictures > 255)
        return EFX_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return EFX_ERR_NO_DEVICE;
    efx_ctx* ctx = new efx_ctx;
    ctx->cfg = *cfg;
    if (ctx->cfg.ring_depth < 2)
        ctx->cfg.ring_depth = 2;
    if (ctx->cfg.max_stream_bytes == 0)
        ctx->cfg.max_stream_bytes = (size_t)16384 * cfg->max_pictures * cfg->max_streams;
    auto bail = [&](int code) {
        efx_destroy(ctx);
        return code;
    };
    if (hipSetDevice(cfg->device) != hipSuccess)
        return bail(EFX_ERR_NO_DEVICE);
    if (cfg->hip_stream)
        ctx->stream = (hipStream_t)cfg->hip_stream;
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(EFX_ERR_DEVICE);
        ctx->own_stream = true;
    }
    const size_t n = (size_t)cfg->max_streams, P = (size_t)cfg->max_pictures, D = (size_t)ctx->cfg.ring_depth;
    ctx->es_cap = ctx->cfg.max_stream_bytes + n * (kEsTailBytes + 32) + kEsGuardBytes;
    ctx->es_cap = (ctx->es_cap + 255) & ~(size_t)255;
    if (ctx->es_cap * kCoefsPerEsByte >= 0xFFFFFFFFull)
        return bail(EFX_ERR_CAPACITY);  // coefficient indices are 32-bit
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) {
        if (e == hipSuccess)
            e = r;
    };
    A(dalloc(&ctx->d_es, ctx->es_cap));
    A(dalloc(&ctx->d_stream_off, n + 1));
    A(dalloc(&ctx->d_stream_perm, n));
    A(dalloc(&ctx->d_tables, 1));
    for (auto& sl : ctx->slot) {
        A(dalloc(&sl.d_pic_count, n));
        A(dalloc(&sl.d_status, n));
        A(dalloc(&sl.d_counters, 1));
        A(dalloc(&sl.d_mbrecs, n * P * kMbCount));
        A(dalloc(&sl.d_coefs, ctx->es_cap * kCoefsPerEsByte));
        A(dalloc(&sl.d_pics, n * P));
        A(dalloc(&sl.d_slices_tmp, n * P * kMaxSlicesPerPicture));
        A(dalloc(&sl.d_qtab, n * P * 64));
        A(dalloc(&sl.d_slice_base, n * P + 1));
        A(dalloc(&sl.d_descs, n * P * kMaxSlicesPerPicture));
    }
    {
        // the parse kernel is a few thousand long-running waves: give it the higher priority so its
        // workgroups are placed as soon as the reconstruction kernels of the previous call free a slot
        int lo = 0, hi = 0;
        A(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (auto& ps : ctx->parse_streams)
            A(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi));
    }
    A(dalloc(&ctx->d_frames, n * D * kFrameBytes + 8192));  // slack: k_recon's window rows may over-read the last frame
    A(dalloc(&ctx->d_video[0], 1));
    A(dalloc(&ctx->d_video[1], 1));
    A(dalloc(&ctx->d_video_lines[0], 1));
    A(dalloc(&ctx->d_video_lines[1], 1));
    A(dalloc(&ctx->d_hash, n * D));
    A(dalloc(&ctx->d_sbc_tables, 1));
    A(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_es), ctx->es_cap, hipHostMallocDefault));
    if (e != hipSuccess) {
        fprintf(stderr, "efx_create: %s\n", hipGetErrorString(e));
        return bail(EFX_ERR_DEVICE);
    }
    ParseTables* pt = new ParseTables;
    build_parse_tables(pt);
    A(hipMemcpy(ctx->d_tables, pt, sizeof(ParseTables), hipMemcpyHostToDevice));
    delete pt;
    for (int ntsc = 0; ntsc < 2; ntsc++) {
        VideoTables vt;
        build_video_tables(ntsc, &vt);
        A(hipMemcpy(ctx->d_video[ntsc], &vt, sizeof(vt), hipMemcpyHostToDevice));
        VideoLineTemplates* lt = new VideoLineTemplates;
        build_video_line_templates(&vt, lt);
        A(hipMemcpy(ctx->d_video_lines[ntsc], lt, sizeof(*lt), hipMemcpyHostToDevice));
        delete lt;
    }
    {
        SbcTables st;
        build_sbc_tables(&st);
        A(hipMemcpy(ctx->d_sbc_tables, &st, sizeof(st), hipMemcpyHostToDevice));
    }
    A(hipMemset(ctx->d_frames, 0, n * D * kFrameBytes));
    A(hipMemset(ctx->d_es, 0, ctx->es_cap));
    for (auto& sl : ctx->slot) {
        A(hipMemset(sl.d_mbrecs, 0, n * P * kMbCount * sizeof(MbRec)));
        A(hipEventCreateWithFlags(&sl.parse_done, hipEventDisableTiming));
        A(hipEventCreateWithFlags(&sl.recon_done, hipEventDisableTiming));
    }
    if (e != hipSuccess)
        return bail(EFX_ERR_DEVICE);
    *out = ctx;
    return EFX_OK;
}

void efx_destroy(efx_ctx* ctx)
{
    if (!ctx)
        return;
    for (auto ps : ctx->parse_streams)
        if (ps)
            (void)hipStreamSynchronize(ps);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    void* bufs[] = {ctx->d_es,   ctx->d_stream_off, ctx->d_stream_perm, ctx->d_tables,
                    ctx->d_frames, ctx->d_video[0],  ctx->d_video[1], ctx->d_video_lines[0], ctx->d_video_lines[1], ctx->d_hash,
                    ctx->d_ts, ctx->d_ts_len, ctx->d_pkt_base, ctx->d_es_len, ctx->d_pes_count, ctx->d_pes, ctx->d_sbc_tables, ctx->d_idx_info, ctx->d_ts_off, ctx->d_idx_len, ctx->d_idx_base, ctx->d_idx_seq};
    for (auto& ev : ctx->ev_demux)
        if (ev)
            (void)hipEventDestroy(ev);
    for (auto& te : ctx->timing_ring)
        for (auto& ev : te.ev)
            if (ev)
                (void)hipEventDestroy(ev);
    for (void* b : bufs)
        if (b)
            (void)hipFree(b);
    for (auto& sl : ctx->slot) {
        void* sb[] = {sl.d_pic_count, sl.d_status, sl.d_counters, sl.d_mbrecs, sl.d_coefs, sl.d_pts,
                      sl.d_pics,      sl.d_slices_tmp, sl.d_qtab, sl.d_slice_base, sl.d_descs};
        for (void* b : sb)
            if (b)
                (void)hipFree(b);
        if (sl.parse_done)
            (void)hipEventDestroy(sl.parse_done);
        if (sl.recon_done)
            (void)hipEventDestroy(sl.recon_done);
    }
    if (ctx->h_es)
        (void)hipHostFree(ctx->h_es);
    for (auto ps : ctx->parse_streams)
        if (ps)
            (void)hipStreamDestroy(ps);
    if (ctx->own_stream && ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}


// transport-stream staging (first EFX_FORMAT_TS upload or first index call): the TS of stream i
// occupies the same region of d_ts that its elementary stream will occupy in d_es (an ES is never
// longer than its TS)
static int ensure_ts_buffers(efx_ctx* ctx)
{
    if (ctx->d_ts)
        return EFX_OK;
    const size_t n_max = (size_t)ctx->cfg.max_streams;
    ctx->pes_cap = ctx->es_cap / 188 + n_max;
    hipError_t e = dalloc(&ctx->d_ts, ctx->es_cap);
    if (e == hipSuccess) e = dalloc(&ctx->d_ts_len, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_pkt_base, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_es_len, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_pes_count, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_pes, ctx->pes_cap);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_info, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_ts_off, n_max + 1);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_len, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_base, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_seq, ctx->pes_cap);
    for (auto& sl : ctx->slot)
        if (e == hipSuccess) e = dalloc(&sl.d_pts, n_max * (size_t)ctx->cfg.max_pictures);
    for (auto& ev : ctx->ev_demux)
        if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e != hipSuccess)
        return fail(ctx, EFX_ERR_DEVICE, "transport-stream buffers", e);
    return EFX_OK;
}

int efx_upload_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* data, const size_t* len, int format)
{
    if (!ctx || !data || !len || n_streams <= 0 || (format != EFX_FORMAT_ES && format != EFX_FORMAT_TS))
        return fail(ctx, EFX_ERR_ARG, "efx_upload_streams: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: more streams than max_streams");
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));  // the bitstream buffer may still be in use
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    const bool is_ts = format == EFX_FORMAT_TS;
    if (is_ts) {
        int r = ensure_ts_buffers(ctx);
        if (r)
            return r;
    }
    static const uint8_t tail[kEsTailBytes] = {0, 0, 0, 1, 0xB7, 0, 0, 1, 0xB7};
    ctx->h_stream_off.assign((size_t)n_streams + 1, 0);
    ctx->h_es_len.assign((size_t)n_streams, 0);
    std::vector<uint32_t> ts_len, pkt_base;
    if (is_ts) {
        ts_len.resize(n_streams);
        pkt_base.resize(n_streams);
    }
    size_t pos = 0, packets = 0;
    for (int i = 0; i < n_streams; i++) {
        const uint8_t* src = data[i];
        const size_t n = len[i];
        if (!src && n)
            return fail(ctx, EFX_ERR_ARG, "efx_upload_streams: null stream");
        const size_t padded = (n + kEsTailBytes + 15) & ~(size_t)15;
        if (pos + padded + kEsGuardBytes > ctx->es_cap)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: more bytes than max_stream_bytes");
        ctx->h_stream_off[i] = pos;
        ctx->h_es_len[i] = (uint32_t)n;
        if (is_ts) {
            ts_len[i] = (uint32_t)n;
            pkt_base[i] = (uint32_t)packets;
            packets += n / 188;
        }
        pos += padded;
    }
    ctx->h_stream_off[n_streams] = pos;
    // Stage into the pinned buffer and ship it: the streams are cut into groups, one host thread
    // copies each group, and a group's H2D transfer is queued as soon as its copy is done, so the
    // transfer of the first groups runs under the copies of the last.
    hipStream_t st = ctx->stream;
    uint8_t* d_dst = is_ts ? ctx->d_ts : ctx->d_es;
    auto stage = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            uint8_t* at = ctx->h_es + ctx->h_stream_off[i];
            const size_t n = len[i], padded = (size_t)(ctx->h_stream_off[i + 1] - ctx->h_stream_off[i]);
            if (n)
                memcpy(at, data[i], n);
            if (is_ts)  // raw packets; k_demux writes the ES, the end-of-data tail and the zero fill
                memset(at + n, 0, padded - n);
            else {
                memcpy(at + n, tail, kEsTailBytes);
                memset(at + n + kEsTailBytes, 0, padded - n - kEsTailBytes);
            }
        }
    };
    {
        unsigned hw = std::thread::hardware_concurrency();
        int groups = pos > ((size_t)4 << 20) ? (int)std::min<unsigned>(8u, std::max(1u, hw / 2)) : 1;
        groups = std::min(groups, n_streams);
        std::vector<std::thread> workers;
        std::vector<int> first((size_t)groups + 1);
        for (int g = 0; g <= groups; g++)
            first[g] = (int)((int64_t)n_streams * g / groups);
        int started = 0;  // groups 1 .. started have a worker; the rest are copied inline (no exception leaves the C-ABI)
        try {
            for (int g = 1; g < groups; g++) {
                workers.emplace_back(stage, first[g], first[g + 1]);
                started = g;
            }
        } catch (...) {
        }
        hipError_t e = hipSuccess;
        for (int g = 0; g < groups; g++) {
            if (g >= 1 && g <= started)
                workers[g - 1].join();
            else
                stage(first[g], first[g + 1]);
            const size_t a = ctx->h_stream_off[first[g]], b = ctx->h_stream_off[first[g + 1]];
            if (e == hipSuccess && b > a)
                e = hipMemcpyAsync(d_dst + a, ctx->h_es + a, b - a, hipMemcpyHostToDevice, st);
        }
        if (e != hipSuccess)
            return fail(ctx, EFX_ERR_DEVICE, "efx_upload_streams: H2D", e);
    }
    // slices of one picture index are dealt to the parse waves stream by stream: longest streams
    // first, so that a wave's 64 slices have similar bit rates (and the long waves start early)
    std::vector<uint32_t> perm((size_t)n_streams);
    for (int i = 0; i < n_streams; i++)
        perm[i] = (uint32_t)i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return len[a] > len[b]; });
    ctx->es_used = pos;
    ctx->n_streams = n_streams;
    ctx->ts_input = is_ts;
    ctx->demux_ms = 0.f;
    ctx->ts_bytes = 0;
    EFX_HIP(hipMemcpyAsync(ctx->d_stream_off, ctx->h_stream_off.data(), ((size_t)n_streams + 1) * sizeof(uint64_t),
                           hipMemcpyHostToDevice, st));
    EFX_HIP(hipMemcpyAsync(ctx->d_stream_perm, perm.data(), (size_t)n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (is_ts) {
        if (packets > ctx->pes_cap)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: PES list capacity");
        EFX_HIP(hipMemsetAsync(ctx->d_ts + pos, 0, kEsGuardBytes, st));
        EFX_HIP(hipMemcpyAsync(ctx->d_ts_len, ts_len.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        EFX_HIP(hipMemcpyAsync(ctx->d_pkt_base, pkt_base.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        EFX_HIP(hipMemsetAsync(ctx->d_es + pos, 0, kEsGuardBytes, st));
        // MpegDecoder::more()/demux() for the whole batch (player.cpp:381-493)
        if (ctx->timing)
            EFX_HIP(hipEventRecord(ctx->ev_demux[0], st));
        hipLaunchKernelGGL(k_demux, dim3(n_streams), dim3(256), 0, st, ctx->d_ts, ctx->d_stream_off, ctx->d_ts_len,
                           ctx->d_pkt_base, ctx->d_es, ctx->d_es_len, ctx->d_pes, ctx->d_pes_count);
        if (ctx->timing)
            EFX_HIP(hipEventRecord(ctx->ev_demux[1], st));
        EFX_HIP(hipGetLastError());
        for (int i = 0; i < n_streams; i++)
            ctx->ts_bytes += ts_len[i];
    } else
        EFX_HIP(hipMemsetAsync(ctx->d_es + pos, 0, kEsGuardBytes, st));
    EFX_HIP(hipStreamSynchronize(st));
    if (is_ts && ctx->timing)
        EFX_HIP(hipEventElapsedTime(&ctx->demux_ms, ctx->ev_demux[0], ctx->ev_demux[1]));
    ctx->uploaded = true;
    ctx->decoded = false;
    ctx->results_valid = false;
    return EFX_OK;
}

int efx_download_es(efx_ctx* ctx, int stream, uint8_t* dst, size_t cap, size_t* es_len)
{
    if (!ctx || !es_len || stream < 0 || stream >= ctx->n_streams || (!dst && cap))
        return EFX_ERR_ARG;
    if (!ctx->uploaded)
        return fail(ctx, EFX_ERR_STATE, "efx_download_es: no streams uploaded");
    size_t n;
    if (ctx->ts_input) {
        uint32_t v = 0;
        EFX_HIP(hipMemcpy(&v, ctx->d_es_len + stream, sizeof(v), hipMemcpyDeviceToHost));
        n = v;
    } else
        n = ctx->h_es_len[stream];
    *es_len = n;
    size_t c = n < cap ? n : cap;
    if (c)
        EFX_HIP(hipMemcpy(dst, ctx->d_es + ctx->h_stream_off[stream], c, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_reset(efx_ctx* ctx)
{
    if (!ctx)
        return EFX_ERR_ARG;
    size_t bytes = (size_t)ctx->cfg.max_streams * ctx->cfg.ring_depth * kFrameBytes;
    EFX_HIP(hipMemsetAsync(ctx->d_frames, 0, bytes, ctx->stream));
    return EFX_OK;
}

int efx_erase_frames(efx_ctx* ctx)
{
    if (!ctx)
        return EFX_ERR_ARG;
    size_t bytes = (size_t)ctx->cfg.max_streams * ctx->cfg.ring_depth * kFrameBytes;
    EFX_HIP(hipMemsetAsync(ctx->d_frames, 0x30, bytes, ctx->stream));
    return EFX_OK;
}

int efx_decode(efx_ctx* ctx)
{
    if (!ctx)
        return EFX_ERR_ARG;
    if (!ctx->uploaded)
        return fail(ctx, EFX_ERR_STATE, "efx_decode: no streams uploaded");
    const int n = ctx->n_streams, P = ctx->cfg.max_pictures, D = ctx->cfg.ring_depth;
    hipStream_t sp = ctx->parse_streams[ctx->calls % kParseStreams], sr = ctx->stream;
    ctx->cur = (int)(ctx->calls++ % kSlots);
    efx_ctx::Slot& sl = ctx->slot[ctx->cur];

    // ---- parse half (parse stream): index -> slice list -> VLC parse + dequantisation ---------------
    EFX_HIP(hipStreamWaitEvent(sp, sl.recon_done, 0));  // the call kSlots back has released this slot
    // macroblock records carry the epoch that wrote them; recycle the tag space by clearing
    if (++sl.epoch > 255) {
        EFX_HIP(hipMemsetAsync(sl.d_mbrecs, 0, (size_t)ctx->cfg.max_streams * P * kMbCount * sizeof(MbRec), sp));
        sl.epoch = 1;
    }
    efx_ctx::TimingEvents* te = nullptr;
    if (ctx->timing && !ctx->timing_ring.empty())
        te = &ctx->timing_ring[ctx->timed_calls++ % kTimingRing];
    if (te)
        EFX_HIP(hipEventRecord(te->ev[0], sp));
    hipLaunchKernelGGL(k_index, dim3(n), dim3(64), 0, sp, ctx->d_es, ctx->d_stream_off, P, sl.d_pics, sl.d_slices_tmp,
                       sl.d_pic_count, sl.d_status, sl.d_qtab, ctx->d_tables->scan, ctx->d_pes, ctx->d_pkt_base,
                       ctx->d_pes_count, ctx->ts_input ? sl.d_pts : nullptr);
    hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, sp, sl.d_pics, sl.d_pic_count, n, P, ctx->d_stream_perm, sl.d_slice_base,
                       sl.d_counters);
    hipLaunchKernelGGL(k_slice_emit, dim3((n * P * kMaxSlicesPerPicture + 255) / 256), dim3(256), 0, sp, sl.d_pics,
                       sl.d_slices_tmp, sl.d_pic_count, ctx->d_stream_off, sl.d_slice_base, n, P, ctx->d_stream_perm, sl.d_descs);
    if (te)
        EFX_HIP(hipEventRecord(te->ev[1], sp));
    const int max_slices = n * P * kMaxSlicesPerPicture;
    hipLaunchKernelGGL(k_parse, dim3((max_slices + 255) / 256), dim3(256), 0, sp, ctx->d_es, sl.d_descs, sl.d_counters,
                       ctx->d_tables, sl.d_mbrecs, sl.d_coefs, sl.d_status, P, sl.epoch);
    if (te)
        EFX_HIP(hipEventRecord(te->ev[2], sp));
    EFX_HIP(hipEventRecord(sl.parse_done, sp));

    // ---- reconstruction half (context stream): one launch per picture index ------------------------------
    EFX_HIP(hipStreamWaitEvent(sr, sl.parse_done, 0));
    if (te)
        EFX_HIP(hipEventRecord(te->ev[4], sr));
    for (int p = 0; p < P; p++)
        hipLaunchKernelGGL(k_recon, dim3(n, (kMbCount * 6 + 63) / 64), dim3(64), 0, sr, sl.d_mbrecs, sl.d_coefs, ctx->d_tables->scan,
                           sl.d_qtab, ctx->d_frames, P, D, p, (p + 1) % D, p % D, sl.epoch);
    if (te)
        EFX_HIP(hipEventRecord(te->ev[3], sr));
    EFX_HIP(hipEventRecord(sl.recon_done, sr));
    EFX_HIP(hipGetLastError());
    ctx->decoded = true;
    ctx->results_valid = false;
    return EFX_OK;
}

int efx_sync(efx_ctx* ctx)
{
    if (!ctx)
        return EFX_ERR_ARG;
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    return EFX_OK;
}

static int fetch_results(efx_ctx* ctx)
{
    if (!ctx->decoded)
        return fail(ctx, EFX_ERR_STATE, "no decode has run");
    if (ctx->results_valid)
        return EFX_OK;
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    const efx_ctx::Slot& sl = ctx->slot[ctx->cur];
    ctx->h_pic_count.resize(ctx->n_streams);
    ctx->h_status.resize(ctx->n_streams);
    EFX_HIP(hipMemcpy(ctx->h_pic_count.data(), sl.d_pic_count, ctx->n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost));
    EFX_HIP(hipMemcpy(ctx->h_status.data(), sl.d_status, ctx->n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost));
    EFX_HIP(hipMemcpy(&ctx->h_counters, sl.d_counters, sizeof(DecodeCounters), hipMemcpyDeviceToHost));
    if (ctx->ts_input) {
        ctx->h_pts.resize((size_t)ctx->n_streams * ctx->cfg.max_pictures);
        EFX_HIP(hipMemcpy(ctx->h_pts.data(), sl.d_pts, ctx->h_pts.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    ctx->results_valid = true;
    return EFX_OK;
}

int efx_picture_count(efx_ctx* ctx, int stream, int* n_pictures)
{
    if (!ctx || !n_pictures || stream < 0 || stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    *n_pictures = (int)ctx->h_pic_count[stream];
    return EFX_OK;
}

int efx_stream_status(efx_ctx* ctx, int stream, uint32_t* bits)
{
    if (!ctx || !bits || stream < 0 || stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    *bits = ctx->h_status[stream];
    return EFX_OK;
}

int efx_picture_pts(efx_ctx* ctx, int stream, int picture, int64_t* pts)
{
    if (!ctx || !pts || stream < 0 || stream >= ctx->n_streams || picture < 0)
        return EFX_ERR_ARG;
    if (!ctx->ts_input) {
        *pts = picture;  // elementary-stream input carries no PTS: pictures are numbered
        return EFX_OK;
    }
    int r = fetch_results(ctx);
    if (r)
        return r;
    *pts = picture < (int)ctx->h_pic_count[stream] ? ctx->h_pts[(size_t)stream * ctx->cfg.max_pictures + picture] : -1;
    return EFX_OK;
}

int efx_picture_slot(const efx_ctx* ctx, int picture)
{
    if (!ctx || picture < 0)
        return EFX_ERR_ARG;
    return (picture + 1) % ctx->cfg.ring_depth;
}

int efx_frame_device_ptr(efx_ctx* ctx, int stream, int slot, void** dptr)
{
    if (!ctx || !dptr || stream < 0 || stream >= ctx->cfg.max_streams || slot < 0 || slot >= ctx->cfg.ring_depth)
        return EFX_ERR_ARG;
    *dptr = ctx->d_frames + ((size_t)stream * ctx->cfg.ring_depth + slot) * kFrameBytes;
    return EFX_OK;
}

int efx_download_frame(efx_ctx* ctx, int stream, int slot, uint8_t* dst)
{
    void* p;
    int r = efx_frame_device_ptr(ctx, stream, slot, &p);
    if (r || !dst)
        return r ? r : EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    EFX_HIP(hipMemcpy(dst, p, kFrameBytes, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_upload_frame(efx_ctx* ctx, int stream, int slot, const uint8_t* src)
{
    void* p;
    int r = efx_frame_device_ptr(ctx, stream, slot, &p);
    if (r || !src)
        return r ? r : EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    EFX_HIP(hipMemcpy(p, src, kFrameBytes, hipMemcpyHostToDevice));
    return EFX_OK;
}

int efx_frame_hashes(efx_ctx* ctx, int first_stream, int n, uint64_t* out)
{
    if (!ctx || !out || first_stream < 0 || n <= 0 || first_stream + n > ctx->cfg.max_streams)
        return EFX_ERR_ARG;
    const int D = ctx->cfg.ring_depth;
    const int frames = n * D;
    hipLaunchKernelGGL(k_frame_hash, dim3((frames + 63) / 64), dim3(64), 0, ctx->stream,
                       ctx->d_frames + (size_t)first_stream * D * kFrameBytes, frames, ctx->d_hash);
    EFX_HIP(hipGetLastError());
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    EFX_HIP(hipMemcpy(out, ctx->d_hash, (size_t)frames * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_video_get_params(int ntsc, efx_video_params* out)
{
    if (!out)
        return EFX_ERR_ARG;
    VideoTables vt;
    build_video_tables(ntsc ? 1 : 0, &vt);
    out->line_width = vt.line_width;
    out->line_count = vt.line_count;
    out->hsync = vt.hsync;
    out->hsync_long = vt.hsync_long;
    out->hsync_short = vt.hsync_short;
    out->burst_start = vt.burst_start;
    out->burst_width = vt.burst_width;
    out->active_start = vt.active_start;
    return EFX_OK;
}

int efx_composite_fields_ex(efx_ctx* ctx, const efx_field_opts* o, uint16_t* dst_device)
{
    if (!ctx || !o || !dst_device || o->first_stream < 0 || o->n_streams <= 0 ||
        o->first_stream + o->n_streams > ctx->cfg.max_streams || o->slot < 0 || o->slot >= ctx->cfg.ring_depth)
        return EFX_ERR_ARG;
    if (o->hscroll && (o->other_slot < 0 || o->other_slot >= ctx->cfg.ring_depth || (o->hscroll & 7) ||
                       o->hscroll <= -EFX_FRAME_WIDTH || o->hscroll >= EFX_FRAME_WIDTH))
        return fail(ctx, EFX_ERR_ARG, "efx_composite_fields_ex: hscroll must be a multiple of 8 in (-352, 352)");
    if (o->overlay_blend < -1)
        return fail(ctx, EFX_ERR_ARG, "efx_composite_fields_ex: overlay_blend must be -1 or >= 0");
    FieldArgs a{};
    a.first_stream = o->first_stream;
    a.ring_depth = ctx->cfg.ring_depth;
    a.slot = o->slot;
    a.other_slot = o->hscroll ? o->other_slot : o->slot;
    a.frame_counter = o->frame_counter;
    a.hscroll = o->hscroll;
    a.overlay = o->overlay;
    a.overlay_stride = o->overlay_stride;
    // composite(), video.cpp:857-859
    int scale = 0;
    if (o->overlay_blend) {
        scale = 255 / 4;
        if (o->overlay_blend != -1 && o->overlay_blend < 32)
            scale = (scale * o->overlay_blend) >> 5;
    }
    a.overlay_scale = scale;
    a.overlay_progress = o->overlay_progress;
    const int lines = o->ntsc ? 262 : 312;
    const int blocks = o->n_streams * ((lines + kCompositeLinesPerBlock - 1) / kCompositeLinesPerBlock);
    hipLaunchKernelGGL(k_composite, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_frames, ctx->d_video[o->ntsc ? 1 : 0],
                       ctx->d_video_lines[o->ntsc ? 1 : 0], a, dst_device);
    EFX_HIP(hipGetLastError());
    return EFX_OK;
}

int efx_composite_fields(efx_ctx* ctx, int first_stream, int n_streams, int slot, int ntsc, int frame_counter,
                         uint16_t* dst_device)
{
    efx_field_opts o{};
    o.first_stream = first_stream;
    o.n_streams = n_streams;
    o.slot = o.other_slot = slot;
    o.ntsc = ntsc;
    o.frame_counter = frame_counter;
    return efx_composite_fields_ex(ctx, &o, dst_device);
}

int efx_pdm(efx_ctx* ctx, int n_streams, const int16_t* pcm_device, int n_samples, int32_t* state_device, uint16_t* dst_device)
{
    if (!ctx || !pcm_device || !state_device || !dst_device || n_streams <= 0 || n_samples <= 0)
        return EFX_ERR_ARG;
    hipLaunchKernelGGL(k_pdm, dim3((n_streams + 63) / 64), dim3(64), 0, ctx->stream, pcm_device, n_streams, n_samples,
                       state_device, dst_device);
    EFX_HIP(hipGetLastError());
    return EFX_OK;
}


// ---- audio elementary stream of a batch of transport streams (push_audio's input) ------------------

int efx_demux_audio(efx_ctx* ctx, int n_streams, const uint8_t* const* ts, const size_t* len, uint8_t* audio_device,
                    size_t stride, uint32_t* audio_len_device)
{
    if (!ctx || !ts || !len || !audio_device || !audio_len_device || n_streams <= 0)
        return fail(ctx, EFX_ERR_ARG, "efx_demux_audio: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_demux_audio: more streams than max_streams");
    int r = ensure_ts_buffers(ctx);
    if (r)
        return r;
    // the TS staging buffers are shared with efx_upload_streams, which only uses them during the call
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> off((size_t)n_streams + 1), out_off((size_t)n_streams + 1);
    std::vector<uint32_t> tlen(n_streams);
    size_t pos = 0;
    for (int i = 0; i < n_streams; i++) {
        if (!ts[i] && len[i])
            return fail(ctx, EFX_ERR_ARG, "efx_demux_audio: null stream");
        if (len[i] > stride)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_demux_audio: stride smaller than a transport stream");
        const size_t padded = (len[i] + 15) & ~(size_t)15;
        if (pos + padded + kEsGuardBytes > ctx->es_cap)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_demux_audio: more bytes than max_stream_bytes");
        off[i] = pos;
        out_off[i] = (uint64_t)i * stride;
        if (len[i])
            memcpy(ctx->h_es + pos, ts[i], len[i]);
        memset(ctx->h_es + pos + len[i], 0, padded - len[i]);
        tlen[i] = (uint32_t)len[i];
        pos += padded;
    }
    off[n_streams] = pos;
    out_off[n_streams] = (uint64_t)n_streams * stride;
    hipStream_t st = ctx->stream;
    uint64_t* d_out_off = nullptr;
    EFX_HIP(dalloc(&d_out_off, (size_t)n_streams + 1));
    hipError_t e = hipMemcpyAsync(ctx->d_ts, ctx->h_es, pos, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_ts_off, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out_off, out_off.data(), out_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_idx_len, tlen.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_demux_audio, dim3(n_streams), dim3(256), 0, st, ctx->d_ts, ctx->d_ts_off, ctx->d_idx_len,
                           audio_device, d_out_off, audio_len_device);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host staging buffer and d_out_off are free again
    (void)hipFree(d_out_off);
    if (e != hipSuccess)
        return fail(ctx, EFX_ERR_DEVICE, "efx_demux_audio", e);
    return EFX_OK;
}

// ---- trick-play index (indexer/indexer.cpp, espflix.cpp:573-629) -------------------------------------

int efx_index_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* ts, const size_t* len, const uint32_t* trick_speed,
                      uint32_t bin_size, efx_idx_rec* recs, uint32_t* samples, size_t samples_cap)
{
    if (!ctx || !ts || !len || !recs || !samples || n_streams <= 0 || bin_size == 0 || samples_cap == 0)
        return fail(ctx, EFX_ERR_ARG, "efx_index_streams: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_index_streams: more streams than max_streams");
    int r = ensure_ts_buffers(ctx);
    if (r)
        return r;
    // the TS staging buffers are shared with efx_upload_streams, which only uses them during the call
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> off((size_t)n_streams + 1);
    std::vector<uint32_t> tlen(n_streams), base(n_streams);
    size_t pos = 0, packets = 0;
    for (int i = 0; i < n_streams; i++) {
        if (!ts[i] && len[i])
            return fail(ctx, EFX_ERR_ARG, "efx_index_streams: null stream");
        const size_t padded = (len[i] + 15) & ~(size_t)15;
        if (pos + padded + kEsGuardBytes > ctx->es_cap)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_index_streams: more bytes than max_stream_bytes");
        off[i] = pos;
        if (len[i])
            memcpy(ctx->h_es + pos, ts[i], len[i]);
        memset(ctx->h_es + pos + len[i], 0, padded - len[i]);
        tlen[i] = (uint32_t)len[i];
        base[i] = (uint32_t)packets;
        packets += len[i] / 188;
        pos += padded;
    }
    off[n_streams] = pos;
    hipStream_t st = ctx->stream;
    uint32_t* d_samples = nullptr;
    EFX_HIP(dalloc(&d_samples, (size_t)n_streams * samples_cap));
    auto done = [&](int code) {
        (void)hipFree(d_samples);
        return code;
    };
    hipError_t e = hipMemcpyAsync(ctx->d_ts, ctx->h_es, pos, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_ts_off, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_idx_len, tlen.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_idx_base, base.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_samples, 0, (size_t)n_streams * samples_cap * sizeof(uint32_t), st);
    if (e != hipSuccess)
        return done(fail(ctx, EFX_ERR_DEVICE, "efx_index_streams: upload", e));
    hipLaunchKernelGGL(k_ts_sequences, dim3(n_streams), dim3(256), 0, st, ctx->d_ts, ctx->d_ts_off, ctx->d_idx_len, ctx->d_idx_base,
                       ctx->d_idx_seq, ctx->d_idx_info);
    hipLaunchKernelGGL(k_idx_bins, dim3((unsigned)((samples_cap + 255) / 256), n_streams), dim3(256), 0, st, ctx->d_idx_seq,
                       ctx->d_idx_base, ctx->d_idx_info, bin_size, d_samples, samples_cap);
    std::vector<IdxInfo> info(n_streams);
    e = hipMemcpyAsync(info.data(), ctx->d_idx_info, n_streams * sizeof(IdxInfo), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(samples, d_samples, (size_t)n_streams * samples_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return done(fail(ctx, EFX_ERR_DEVICE, "efx_index_streams: kernels", e));
    int status = EFX_OK;
    for (int i = 0; i < n_streams; i++) {
        efx_idx_rec& rec = recs[i];
        memset(&rec, 0, sizeof(rec));
        rec.first_pts = info[i].first_pts;
        rec.last_pts = info[i].last_pts;
        rec.bin_size = bin_size;
        rec.trick_speed = trick_speed ? trick_speed[i] : 1;
        const int64_t end = rec.last_pts - rec.first_pts;
        if (!info[i].n_seq || end < 0)
            continue;  // no sequence header: sample_count 0 (the indexer would index out of an empty list)
        const uint64_t count = (uint64_t)(end / bin_size) + 1;
        if (count > samples_cap) {
            status = fail(ctx, EFX_ERR_CAPACITY, "efx_index_streams: samples_cap too small");
            continue;
        }
        rec.sample_count = (uint32_t)count;
    }
    return done(status);
}

size_t efx_idx_build(const efx_idx_rec recs[3], const uint32_t* const samples[3], uint8_t* out, size_t cap)
{
    if (!recs || !samples)
        return 0;
    size_t total = 8 + 3 * sizeof(efx_idx_rec);
    for (int k = 0; k < 3; k++)
        total += 4 * (size_t)recs[k].sample_count;
    if (!out || total > cap)
        return total;
    const uint32_t sig = 'I' | ('D' << 8) | ('X' << 16), three = 3;  // merge_index, indexer.cpp:225-227
    memcpy(out, &sig, 4);
    memcpy(out + 4, &three, 4);
    uint8_t* p = out + 8;
    for (int k = 0; k < 3; k++) {
        efx_idx_rec r = recs[k];
        r.reserved = 0;
        memcpy(p, &r, sizeof(r));
        p += sizeof(r);
    }
    for (int k = 0; k < 3; k++) {
        memcpy(p, samples[k], 4 * (size_t)recs[k].sample_count);
        p += 4 * (size_t)recs[k].sample_count;
    }
    return total;
}

static void idx_recs(const void* hdr, efx_idx_rec r[3]) { memcpy(r, static_cast<const uint8_t*>(hdr) + 8, 3 * sizeof(efx_idx_rec)); }

static int64_t idx_map_pts(int64_t pts, const efx_idx_rec& r, const efx_idx_rec& video)  // espflix.cpp:589-594
{
    pts -= r.first_pts;
    pts *= video.last_pts - video.first_pts;
    const int64_t d = r.last_pts - r.first_pts;
    return d ? pts / d : 0;  // a one-picture trick stream would divide by zero in the reference
}

int64_t efx_idx_pts2pts(const void* idx_hdr, int64_t pts, int speed)  // espflix.cpp:597-604
{
    efx_idx_rec r[3];
    idx_recs(idx_hdr, r);
    if (speed == 1)
        return r[0].first_pts + idx_map_pts(pts, r[1], r[0]);
    if (speed == -1)
        return r[0].last_pts - idx_map_pts(pts, r[2], r[0]);
    return pts;
}

uint32_t efx_idx_pts2offset(const void* idx_hdr, int64_t pts, int speed)  // espflix.cpp:607-627
{
    efx_idx_rec r[3];
    idx_recs(idx_hdr, r);
    const efx_idx_rec &video = r[0], &fwd = r[1], &rwd = r[2];
    pts = std::max(std::min(pts, video.last_pts), video.first_pts);
    uint32_t offset;
    switch (speed) {
    case 1:
        offset = (uint32_t)(pts - video.first_pts) / fwd.trick_speed / fwd.bin_size;
        offset = std::min(fwd.sample_count - 1, offset);
        offset += video.sample_count;
        break;
    case -1:
        offset = (uint32_t)((video.last_pts - pts) - video.first_pts) / rwd.trick_speed / rwd.bin_size;
        offset = std::min(rwd.sample_count - 1, offset);
        offset += video.sample_count + fwd.sample_count;
        break;
    default:
        offset = (uint32_t)((pts - video.first_pts) / video.bin_size);
        offset = std::min(video.sample_count - 1, offset);
        break;
    }
    return offset * 4 + (uint32_t)(8 + 3 * sizeof(efx_idx_rec));
}

size_t efx_sbc_state_bytes(void) { return sizeof(SbcState); }

int efx_sbc_decode(efx_ctx* ctx, int n_streams, const uint8_t* frames_device, size_t stream_stride, int frame_bytes,
                   int n_frames, void* state_device, int16_t* pcm_device, size_t pcm_stride, uint32_t* ret_device,
                   uint32_t* pcm_count_device, int flags)
{
    if (!ctx || !frames_device || !state_device || !pcm_device || n_streams <= 0 || n_frames < 0 || frame_bytes <= 0 ||
        (size_t)frame_bytes * (size_t)n_frames > 0x7FFFFFFFu || ((uintptr_t)state_device & 3))
        return EFX_ERR_ARG;
    hipLaunchKernelGGL(k_sbc, dim3(n_streams), dim3(64), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames,
                       static_cast<SbcState*>(state_device), ctx->d_sbc_tables, pcm_device, pcm_stride, ret_device,
                       pcm_count_device, flags);
    EFX_HIP(hipGetLastError());
    return EFX_OK;
}

int efx_set_timing(efx_ctx* ctx, int enable)
{
    if (!ctx)
        return EFX_ERR_ARG;
    if (enable && ctx->timing_ring.empty()) {
        ctx->timing_ring.resize(kTimingRing);
        for (auto& te : ctx->timing_ring)
            for (auto& ev : te.ev)
                EFX_HIP(hipEventCreate(&ev));
    }
    ctx->timing = enable != 0;
    ctx->timed_calls = 0;  // (re-)enabling starts a new averaging window
    return EFX_OK;
}

int efx_get_timing(efx_ctx* ctx, efx_timing* out)
{
    if (!ctx || !out)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    efx_timing t{};
    // mean over the efx_decode calls since efx_set_timing(1) (the last kTimingRing of them);
    // fetch_results() has synchronised both streams, so every recorded event is complete
    const uint64_t n_timed = std::min<uint64_t>(ctx->timed_calls, kTimingRing);
    for (uint64_t k = 0; k < n_timed; k++) {
        const efx_ctx::TimingEvents& te = ctx->timing_ring[(ctx->timed_calls - 1 - k) % kTimingRing];
        float a = 0, b = 0, c = 0, d = 0;
        EFX_HIP(hipEventElapsedTime(&a, te.ev[0], te.ev[1]));
        EFX_HIP(hipEventElapsedTime(&b, te.ev[1], te.ev[2]));
        EFX_HIP(hipEventElapsedTime(&c, te.ev[4], te.ev[3]));
        EFX_HIP(hipEventElapsedTime(&d, te.ev[0], te.ev[3]));
        t.index_ms += a / n_timed;
        t.parse_ms += b / n_timed;
        t.recon_ms += c / n_timed;
        t.total_ms += d / n_timed;
    }
    t.timed_calls = (uint32_t)n_timed;
    for (int i = 0; i < ctx->n_streams; i++)
        t.pictures += ctx->h_pic_count[i];
    t.slices = ctx->h_counters.total_slices;
    t.coefficients = ctx->h_counters.coefficients;
    t.es_bytes = ctx->es_used;
    t.demux_ms = ctx->demux_ms;
    t.ts_bytes = ctx->ts_bytes;
    *out = t;
    return EFX_OK;
}


int efx_device_alloc(efx_ctx* ctx, size_t bytes, void** dptr)
{
    if (!ctx || !dptr)
        return EFX_ERR_ARG;
    EFX_HIP(hipMalloc(dptr, bytes));
    return EFX_OK;
}

int efx_device_free(efx_ctx* ctx, void* dptr)
{
    if (!ctx)
        return EFX_ERR_ARG;
    EFX_HIP(hipFree(dptr));
    return EFX_OK;
}

int efx_memcpy_h2d(efx_ctx* ctx, void* dst_device, const void* src, size_t bytes)
{
    if (!ctx || !dst_device || !src)
        return EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    EFX_HIP(hipMemcpy(dst_device, src, bytes, hipMemcpyHostToDevice));
    return EFX_OK;
}

int efx_memcpy_d2h(efx_ctx* ctx, void* dst, const void* src_device, size_t bytes)
{
    if (!ctx || !dst || !src_device)
        return EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    EFX_HIP(hipMemcpy(dst, src_device, bytes, hipMemcpyDeviceToHost));
    return EFX_OK;
}

}  // extern "C"
