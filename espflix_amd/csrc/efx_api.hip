// efx_api.hip -- the C-ABI of libefx (include/efx.h): context, uploads, kernel launches.
//
// Host orchestration only; all arithmetic of the hot path lives in the k_*.hip kernels.  There
// is no CPU fallback anywhere: without a gfx950 device efx_create fails with
// EFX_ERR_NO_DEVICE / EFX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <thread>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "efx.h"
#include "efx_internal.h"
#include "parse_tm.h"

namespace efx {
// kernels (k_demux.hip, k_index.hip, k_parse.hip, k_recon.hip, k_video.hip)
struct DemuxChunk;
__global__ void k_demux_scan(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, DemuxChunk*);
__global__ void k_demux_offsets(const uint32_t*, const uint32_t*, DemuxChunk*, uint8_t*, const uint64_t*, uint32_t*, uint32_t*);
__global__ void k_demux(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, const DemuxChunk*, uint8_t*, PesEntry*);
__global__ void k_demux_fused(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, DemuxChunk*, uint8_t*, PesEntry*, uint32_t*, uint32_t*,
                              uint32_t*);
__global__ void k_demux_audio_scan(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, DemuxChunk*);
__global__ void k_demux_audio_offsets(const uint32_t*, const uint32_t*, DemuxChunk*, uint32_t*);
__global__ void k_demux_audio(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, const DemuxChunk*, uint8_t*, const uint64_t*);
__global__ void k_ts_sequences(const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, PesEntry*, IdxInfo*);
__global__ void k_idx_bins(const PesEntry*, const uint32_t*, const IdxInfo*, uint32_t, uint32_t*, size_t);
__global__ void k_index(const uint8_t*, const uint64_t*, int, PicInfo*, SliceTmp*, uint32_t*, uint32_t*, uint32_t*,
                        const uint32_t*, const PesEntry*, const uint32_t*, const uint32_t*, int64_t*, int64_t*, int, int, int);
__global__ void k_advance(StreamState*, const uint32_t*, int64_t*, const int64_t*, int, int, int, int32_t*);
__global__ void k_slice_scan(const PicInfo*, const uint32_t*, int, int, const uint32_t*, uint32_t*, DecodeCounters*);
__global__ void k_slice_emit(const PicInfo*, const SliceTmp*, const uint32_t*, const uint64_t*, const uint32_t*, int, int,
                             const uint32_t*, SliceDesc*, uint32_t*);
__global__ void k_parse(const uint8_t*, const SliceDesc*, DecodeCounters*, const TmTables*, MbRec*, TmU4*, uint32_t*, uint32_t*,
                        int, int);
__global__ void k_recon(const MbRec*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*, int, int, int, const int32_t*, int, int);
__global__ void k_recon_all(const MbRec*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*, int, int, int, const int32_t*, int, int,
                            int, uint32_t*, uint32_t*, int);

__global__ void k_frame_hash(const uint8_t*, int, uint64_t*);
__global__ void k_fill(uint32_t*, uint32_t, size_t);
__global__ void k_composite(const uint8_t*, const VideoTables*, const VideoLineTemplates*, FieldArgs, uint16_t*);
__global__ void k_pdm(const int16_t*, int, int, int32_t*, uint16_t*);
__global__ void k_sbc(const uint8_t*, size_t, int, int, SbcState*, const SbcTables*, int16_t*, size_t, uint32_t*, uint32_t*, int);
__global__ void k_sbc_frames(const uint8_t*, size_t, int, int, SbcFrameInfo*, uint32_t*, uint32_t*, SbcQueues*);
__global__ void k_sbc_plan(const SbcFrameInfo*, int, int, const SbcState*, SbcFramePlan*, uint32_t*, SbcQueues*, uint32_t*, int, int, uint32_t*,
                           uint32_t*, SbcExtraItem*, uint8_t*);
__global__ void k_sbc_par_stereo(const uint8_t*, size_t, int, int, const SbcState*, SbcState*, const SbcTables*, const SbcFrameInfo*, int16_t*,
                                 size_t, uint32_t*, int, SbcQueues*, const uint32_t*, int, const SbcExtraItem*, int);
__global__ void k_sbc_par_mono(const uint8_t*, size_t, int, int, const SbcState*, SbcState*, const SbcTables*, const SbcFrameInfo*, int16_t*,
                               size_t, uint32_t*, int, SbcQueues*, const uint32_t*, int, const SbcExtraItem*, int);
__global__ void k_sbc_gen(const uint8_t*, size_t, int, int, const SbcState*, SbcState*, const SbcTables*, const SbcFrameInfo*,
                          const SbcFramePlan*, int16_t*, size_t, int, SbcQueues*, const uint32_t*, int, const uint8_t*, int);
__global__ void k_sbc_finish(const uint8_t*, size_t, int, int, SbcState*, const SbcState*, const SbcTables*, int16_t*, size_t, uint32_t*,
                             uint32_t*, int, uint32_t*);
}  // namespace efx

using namespace efx;

#ifndef EFX_SBC_WG_PER_CU
#define EFX_SBC_WG_PER_CU 4  // k_sbc_par_mono workgroups per compute unit (k_sbc.hip: EFX_SBC_WAVES)
#endif

#ifndef EFX_PARSE_STREAMS
#define EFX_PARSE_STREAMS 2
#endif
#ifndef EFX_SLOTS
#define EFX_SLOTS 3
#endif
constexpr int kParseStreams = EFX_PARSE_STREAMS;  // parse halves in flight at once (latency-bound kernels: two overlap well)
constexpr int kMaxGroups = 16;      // an efx_decode call parses its streams in up to this many parse halves
#ifndef EFX_RECON_MERGE
#define EFX_RECON_MERGE 1
#endif
constexpr int kReconMerge = EFX_RECON_MERGE;  // parse halves whose streams are reconstructed by ONE launch per picture index
#ifndef EFX_GROUP_STREAMS
#define EFX_GROUP_STREAMS 1024
#endif
constexpr int kGroupStreams = EFX_GROUP_STREAMS;
constexpr int kTimingRing = 64;   // efx_decode calls whose stage times efx_get_timing can average
constexpr int kSlots = EFX_SLOTS;         // parse -> recon hand-over buffer sets (one being reconstructed + two being parsed)
constexpr int kUploads = 3;       // bitstream buffers: up to two being parsed (two parse halves are in flight), one being filled --
                                  // with two, an upload had to wait for the parse half two calls back before its transfer could start
// k_recon_all's hand-over words: 64 lines of 128 bytes (queue heads, spins, abort, development statistics), then one line per stream
constexpr size_t kReconSyncWords(size_t streams) { return (64 + streams) * 32; }

struct efx_ctx {
    efx_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool pooled_streams = false;  // the parse / copy (and, if own_stream, reconstruction) streams go back to the pool
    unsigned stream_serial = 0;
    std::string err;

    size_t es_cap = 0;  // bytes of one ES buffer (including tails and guard)
    bool decoded = false, results_valid = false;

    ParseTables* d_tables = nullptr;
    TmTables* d_tm_tables = nullptr;  // the token machine's tables (parse_tm.h)
    uint8_t* d_frames = nullptr;
    StreamState* d_state = nullptr;  // per stream: ring position, "a PTS has been seen", newest PES PTS (k_advance)

    // One resident batch of bitstreams.  efx_upload_streams fills the buffer the last decode does NOT read
    // (staging copy on the host, H2D and k_demux on the copy stream) while the GPU still decodes the other
    // one: ingest and decode of consecutive batches overlap.
    struct Upload {
        uint8_t* d_es = nullptr;
        uint64_t* d_stream_off = nullptr;
        uint32_t* d_stream_perm = nullptr;  // streams by descending length: the order of slices inside a picture index
        // transport-stream input (allocated on the first EFX_FORMAT_TS upload)
        uint32_t* d_ts_len = nullptr;    // per stream: TS bytes
        uint32_t* d_pkt_base = nullptr;  // per stream: first entry of its PES list (= packets before it)
        uint32_t* d_es_len = nullptr;    // per stream: demuxed ES bytes (without tail)
        uint32_t* d_pes_count = nullptr;
        PesEntry* d_pes = nullptr;
        uint8_t* h_es = nullptr;         // pinned staging: the bitstreams, then the small per-stream arrays
        uint8_t* h_meta = nullptr;       // pinned: stream_off (n + 1 x u64) | perm | ts_len | pkt_base (n x u32 each)
        std::vector<uint64_t> stream_off;
        std::vector<uint32_t> es_len;    // per stream, ES input only
        int n_streams = 0;
        int sort_groups = 1;  // d_stream_perm orders the streams by length INSIDE each of this many groups (group_first)
        size_t es_used = 0, ts_bytes = 0;
        bool ts_input = false, valid = false;
        hipEvent_t uploaded = nullptr;                                  // H2D (+ k_demux) done, copy stream
        hipEvent_t last_read[kParseStreams] = {};       // newest parse half that reads this buffer
        hipEvent_t ev_demux[2] = {nullptr, nullptr};
        bool demux_timed = false;
    } up[kUploads];
    int cur_up = -1;  // batch the next efx_decode reads
    // page-locked host arenas the caller lays batches out in (efx_host_alloc / efx_host_register): base, bytes, ours to free
    struct Arena {
        uint8_t* base;
        size_t bytes;
        bool owned;
    };
    std::vector<Arena> arenas;
    hipStream_t copy_stream = nullptr;
    // transient TS staging on the device + the lists of efx_index_streams / efx_demux_audio
    uint8_t* d_ts = nullptr;
    DemuxChunk* d_demux_chunks = nullptr;  // k_demux's per-chunk totals / bases (16 bytes each; one demux runs at a time)
    size_t pes_cap = 0;
    IdxInfo* d_idx_info = nullptr;
    uint64_t* d_ts_off = nullptr;
    uint32_t* d_idx_len = nullptr;
    uint32_t* d_idx_base = nullptr;
    PesEntry* d_idx_seq = nullptr;
    bool ts_ready = false;  // every transport-stream buffer and event exists

    // kSlots sets of parse -> recon hand-over buffers: efx_decode() number n parses into slot
    // n % kSlots on parse stream n % kParseStreams while the recon stream is still reconstructing
    // earlier calls from the other slots.  The parse half is bound by the latency of its longest
    // waves (3 waves per SIMD, ~1/3 of the issue slots used), so back-to-back decodes keep two parse
    // halves and one reconstruction half on the GPU at once.  Every slot carries its own
    // index / slice-list scratch.
    struct Slot {
        uint32_t* d_pic_count = nullptr;
        uint32_t* d_status = nullptr;
        DecodeCounters* d_counters = nullptr;  // kMaxGroups of them: one per group of a call
        MbRec* d_mbrecs = nullptr;
        TmU4* d_raw = nullptr;       // k_parse's own scratch: raw macroblock records, pass 1 -> pass 2 (same shape as d_mbrecs)
        uint32_t* d_coefs = nullptr;
        // index / slice-list scratch of this call (per slot: two parse halves run concurrently)
        PicInfo* d_pics = nullptr;
        SliceTmp* d_slices_tmp = nullptr;
        uint32_t* d_qtab = nullptr;  // per (stream, picture) custom quantiser tables, read by k_recon
        uint32_t* d_slice_base = nullptr;
        SliceDesc* d_descs = nullptr;
        int64_t* d_pts = nullptr;    // per (stream, picture): PTS latched at the picture header (TS input); then, per
                                     // stream, the newest PES PTS of the upload (k_index -> k_advance)
        int32_t* d_call_pos = nullptr;  // per stream: ring position of this call's first picture, first picture with a PTS
        uint32_t* d_recon_sync = nullptr;  // k_recon_all: queue heads, spins, abort, finished items per stream -- a 128-byte line each
        hipEvent_t parse_done[kReconMerge] = {}, recon_done = nullptr, wrap_cleared = nullptr;
        int epoch = 0;
        int upload = 0;  // batch this call decoded
    } slot[kSlots];
    // stage timing: one event set per efx_decode call since efx_set_timing(1), so that a run of
    // back-to-back (overlapping) calls can be averaged afterwards without a host sync in between
    struct TimingEvents {
        // per parse half: parse start, index end, parse end (its parse stream); the first half of a reconstruction group
        // also carries that group's recon end [3] and recon start [4] (context stream)
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    };
    std::vector<TimingEvents> timing_ring;  // kTimingRing x kMaxGroups sets, created by efx_set_timing
    uint64_t timed_calls = 0;               // calls recorded since timing was (re-)enabled
    struct Group {
        int slot, first, count;  // hand-over slot, streams [first, first + count)
        int half0, halves;       // its parse halves (their counters: d_counters[half0 ...])
    };
    Group groups[kMaxGroups];  // reconstruction groups of the most recent efx_decode
    int n_groups = 0;
    hipEvent_t last_recon_done = nullptr;  // end of the most recently queued reconstruction group
    // Reconstruction groups queued since the host last KNEW the reconstruction stream idle (context creation, any call that
    // synchronises it: sync_all).  The launch structure of a decode call is decided from this count, not from a hipEventQuery of
    // the stream (round 5: a race against the GPU -- the same call sequence could run different launches from run to run):
    // 0 = the call finds the stream idle by the library's own book-keeping (it is split into groups, its first parse half may
    // have the whole chip), otherwise it pipelines against what is queued (one group, capped parser).  A caller that lets the
    // GPU drain WITHOUT synchronising is treated as back to back: that costs the one call some latency, never a result.
    uint64_t recon_groups_unsynced = 0;
    int last_upload = 0;       // batch the most recent efx_decode read
    int last_n_streams = 0;    // ... its stream count and format, as they were when the decode was queued (a later upload may
    bool last_ts_input = false;  // have recycled the Upload record by the time the results are fetched)
    uint64_t subcalls = 0;     // reconstruction groups launched so far: group k uses slot k % kSlots
    uint64_t parse_calls = 0;  // parse halves launched so far: half k runs on parse stream k % kParseStreams
    uint8_t timing_groups[kTimingRing] = {};    // parse halves of the timed call
    uint16_t timing_launches[kTimingRing] = {};  // reconstruction kernel launches of the timed call
    uint16_t timing_leaders[kTimingRing] = {};  // bit h: half h is the first of its reconstruction group (carries ev[3], ev[4])
    hipStream_t parse_streams[kParseStreams] = {};
    VideoTables* d_video[2] = {nullptr, nullptr};  // [0] PAL, [1] NTSC
    VideoLineTemplates* d_video_lines[2] = {nullptr, nullptr};
    SbcTables* d_sbc_tables = nullptr;
    int parse_wg_cap = 0;  // k_parse workgroups resident per parse kernel while reconstruction launches are queued (0: no cap); EFX_PARSE_WG_CAP
    // launch structure (efx_set_option; the environment variables of the same names, upper case with EFX_, set the defaults)
    int opt_groups = 0;        // reconstruction groups per call: 0 = one group behind a busy reconstruction stream, groups of
                               // kGroupStreams streams when it is idle; n >= 1: always n
    int opt_parse_cap = 0;     // 0 = cap the parse kernel's residency only while reconstruction is queued; 1 always; 2 never
    int opt_recon_mode = 0;    // 0 = one k_recon launch per picture index (default: measured faster beside the parse halves,
                               // profiles/r5_recon_all.md); 1, 2 = one k_recon_all launch per group
    int opt_recon_waves = 0;   // k_recon_all with opt_recon_items = 0: workgroups per compute unit (0 = 18, what its LDS admits)
    int opt_recon_items = 16;  // k_recon_all: items a wave takes before it ends (0: until none is left)
    int n_cus = 256;
    // efx_sbc_decode's scratch.  d_sbc_flags: n words per stream-indexed array -- [0, n) which kernel decodes the stream
    // (k_sbc_frames / k_sbc_plan), [n, 4n) the three work lists; then the SbcQueues
    uint32_t* d_sbc_flags = nullptr;
    SbcState* d_sbc_next = nullptr;   // the state a frame-parallel kernel leaves, put in place by k_sbc_commit
    size_t sbc_flags_cap = 0;
    SbcFrameInfo* d_sbc_info = nullptr;  // per (stream, frame)
    SbcFramePlan* d_sbc_plan = nullptr;  // per (stream, frame + 1)
    SbcExtraItem* d_sbc_extra = nullptr; // two lists of (stream, granule of eight frames) capacity each
    uint8_t* d_sbc_cover = nullptr;      // per (stream, granule): a regular kernel decodes it
    size_t sbc_info_cap = 0;             // in frames: streams x (frames + 1)
    size_t sbc_gran_cap = 0;             // in granules: streams x ceil(frames / 8)
    int opt_sbc_serial = 0;              // 1 = every stream through k_sbc, one wave per stream (the tests' comparison)
    int opt_demux_fused = 0;             // 1 = TS input through the one-pass demultiplexer (k_demux_fused: measured slower, kept as an option)
    bool sbc_flags_clean = false;        // d_sbc_flags[0, n) all kSbcRegularFlag (k_sbc_finish leaves them so)
    uint64_t* d_hash = nullptr;

    // results of the last decode (fetch_results)
    int n_streams = 0;  // streams of the batch the last decode read
    std::vector<int64_t> h_pts;  // per (stream, picture), TS input only
    std::vector<uint32_t> h_pic_count, h_status;
    std::vector<int32_t> h_call_pos;
    DecodeCounters h_counters{};

    bool timing = false;
};

namespace {

// The streams of a context -- reconstruction, parse, copy -- are kept when the context goes and handed to the next context
// on the same device.  Where a stream's hardware queue sits on the command processor's pipes decides whether the parse
// halves run BESIDE the reconstruction or take turns with it (efx_create), and the runtime deals recycled queues out in
// whatever order they came back: the second context of a process decoded the same batch at 4.8 instead of 8.2 M
// frames/s (tools/efx_scale, round 3).  With the set reused as a whole every later context runs where the first one ran.
struct StreamSet {
    hipStream_t recon = nullptr, parse[kParseStreams] = {}, copy = nullptr;
    unsigned serial = 0;  // creation order: the oldest parked set of a device is handed out first (the first set a process
                          // creates is the one whose queues were dealt in the order efx_create asks for)
};
unsigned g_set_serial = 0;
std::mutex g_pool_guard;
std::vector<std::pair<int, StreamSet>> g_stream_pool;  // (device, parked set)

bool take_stream_set(int device, StreamSet* out)
{
    std::lock_guard<std::mutex> lk(g_pool_guard);
    size_t best = g_stream_pool.size();
    for (size_t i = 0; i < g_stream_pool.size(); i++)
        if (g_stream_pool[i].first == device && (best == g_stream_pool.size() || g_stream_pool[i].second.serial < g_stream_pool[best].second.serial))
            best = i;
    if (best == g_stream_pool.size())
        return false;
    *out = g_stream_pool[best].second;
    g_stream_pool.erase(g_stream_pool.begin() + (ptrdiff_t)best);
    return true;
}
void park_stream_set(int device, const StreamSet& set)
{
    std::lock_guard<std::mutex> lk(g_pool_guard);
    g_stream_pool.emplace_back(device, set);
}

// An efx_decode call that finds the GPU idle runs as G groups of about kGroupStreams streams, one after the other: each
// group is a complete parse half + reconstruction half with its own hand-over slot, so that the parse half of the next
// group runs beside the reconstruction of this one -- the call pipelines inside itself.  Back-to-back calls pipeline
// against each other and run as one group (efx_decode_range).  Streams are independent: results do not depend on G.
int group_count(int n_streams)
{
    int g = (n_streams + kGroupStreams / 2) / kGroupStreams;
    return g < 1 ? 1 : (g > kMaxGroups ? kMaxGroups : g);
}
// streams [first, end) of group g: boundaries on multiples of 8, so that a stream's workgroups stay on one XCD
int group_first(int n_streams, int groups, int g)
{
    return g >= groups ? n_streams : (int)(((int64_t)n_streams * g / groups) & ~7);
}

// A context's device is made current on every entry: efx_multi_context() hands per-device contexts to the caller's own
// thread (composite, PDM, timing, raw allocation), whose current device is whatever it was.
void bind_device(efx_ctx* c);

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

// ---- device memory ------------------------------------------------------------------------------------------------------
// Product: hipMalloc.  With EFX_GUARD=1 (or 2) in the environment every device buffer of the library -- and every buffer
// handed out through efx_device_alloc -- is its own virtual-memory mapping (hipMemAddressReserve / hipMemCreate / hipMemMap)
// that ENDS on the last mapped byte of its range (mode 1; mode 2: starts on the first), with an unmapped page on either
// side: a kernel that reads or writes one byte too far faults at once ("Memory access fault by GPU", the kernel named
// under AMD_SERIALIZE_KERNEL=3) instead of quietly touching whatever hipMalloc's pooling put there.  A debugging aid
// (bench.py --soak, profiles/r5_fault_hunt.md): the kernels' deliberate over-reads -- k_recon's window rows behind the
// last frame, the parse lanes' read-ahead behind the last stream -- stay inside the slack their buffers are allocated with.
struct GuardRec {
    void* va;
    size_t va_bytes, map_bytes, gran;
    hipMemGenericAllocationHandle_t handle;
};
std::mutex g_guard_lock;
std::vector<std::pair<void*, GuardRec>> g_guard_recs;

int guard_mode()
{
    static const int mode = [] {
        const char* e = getenv("EFX_GUARD");
        const int m = e ? atoi(e) : 0;
        if (m)
            fprintf(stderr, "libefx: EFX_GUARD=%d -- every device buffer is its own mapping between two unmapped pages (%s-aligned)\n", m,
                    m == 2 ? "start" : "end");
        return m;
    }();
    return mode;
}

hipError_t guard_alloc(void** out, size_t bytes)
{
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess)
        return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess)
        return e;
    if (bytes == 0)
        bytes = 16;
    GuardRec r{};
    r.gran = gran;
    r.map_bytes = (bytes + gran - 1) / gran * gran;
    r.va_bytes = r.map_bytes + 2 * gran;
    if ((e = hipMemAddressReserve(&r.va, r.va_bytes, 0, nullptr, 0)) != hipSuccess)
        return e;
    if ((e = hipMemCreate(&r.handle, r.map_bytes, &prop, 0)) != hipSuccess) {
        (void)hipMemAddressFree(r.va, r.va_bytes);
        return e;
    }
    char* base = static_cast<char*>(r.va) + gran;
    hipMemAccessDesc ad{};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemMap(base, r.map_bytes, 0, r.handle, 0)) != hipSuccess || (e = hipMemSetAccess(base, r.map_bytes, &ad, 1)) != hipSuccess) {
        (void)hipMemRelease(r.handle);
        (void)hipMemAddressFree(r.va, r.va_bytes);
        return e;
    }
    // mode 1: the buffer ENDS with the mapping, so its start is 16-byte aligned and no more (round-5 ADVICE: the comment here
    // claimed 256): every kernel is correct at 16 bytes -- the widest access is a dwordx4 -- and layouts that are tuned to
    // 128-byte lines (one hand-over word per line in d_recon_sync) merely straddle lines in this debugging mode.  Mode 2
    // (the buffer STARTS with the mapping) keeps the allocator's natural alignment.
    const size_t span = (bytes + 255) & ~(size_t)255;
    void* p = guard_mode() == 2 ? base : base + (r.map_bytes - span) + (span - ((bytes + 15) & ~(size_t)15));
    {
        std::lock_guard<std::mutex> lk(g_guard_lock);
        g_guard_recs.emplace_back(p, r);
    }
    *out = p;
    return hipSuccess;
}

hipError_t dev_alloc(void** p, size_t bytes) { return guard_mode() ? guard_alloc(p, bytes) : hipMalloc(p, bytes); }

hipError_t dev_free(void* p)
{
    if (!p)
        return hipSuccess;
    if (!guard_mode())
        return hipFree(p);
    GuardRec r{};
    {
        std::lock_guard<std::mutex> lk(g_guard_lock);
        size_t i = 0;
        while (i < g_guard_recs.size() && g_guard_recs[i].first != p)
            i++;
        if (i == g_guard_recs.size())
            return hipFree(p);  // (not ours)
        r = g_guard_recs[i].second;
        g_guard_recs.erase(g_guard_recs.begin() + (ptrdiff_t)i);
    }
    (void)hipDeviceSynchronize();  // hipFree's implicit barrier
    hipError_t e = hipMemUnmap(static_cast<char*>(r.va) + r.gran, r.map_bytes);
    hipError_t e2 = hipMemRelease(r.handle);
    // (debugging mode only) The address range is NOT given back, so a long soak run under EFX_GUARD grows its address space
    // by every buffer it ever freed: a freed pointer then stays unmapped for the rest of the process (a use after free
    // faults too), and the runtime never sees an address twice -- one soak process of this round died in the HIP runtime's own
    // bookkeeping ("Memobj map does not have ptr") after some two thousand reserve / free cycles of recycled ranges.
    return e != hipSuccess ? e : e2;
}

template <typename T>
hipError_t dalloc(T** p, size_t n)
{
    return dev_alloc(reinterpret_cast<void**>(p), n * sizeof(T));
}

int sync_all(efx_ctx* ctx)
{
    EFX_HIP(hipStreamSynchronize(ctx->copy_stream));
    for (auto ps : ctx->parse_streams)
        EFX_HIP(hipStreamSynchronize(ps));
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;
    return EFX_OK;
}

size_t meta_bytes(size_t n) { return (n + 1) * sizeof(uint64_t) + 3 * n * sizeof(uint32_t); }

void bind_device(efx_ctx* c)
{
    if (c)
        (void)hipSetDevice(c->cfg.device);
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
    StreamSet pooled;
    if (!cfg->hip_stream && take_stream_set(cfg->device, &pooled)) {
        // the stream set of an earlier context on this device: same queues, same places
        ctx->stream = pooled.recon;
        ctx->own_stream = true;
        for (int i = 0; i < kParseStreams; i++)
            ctx->parse_streams[i] = pooled.parse[i];
        ctx->copy_stream = pooled.copy;
        ctx->pooled_streams = true;
        ctx->stream_serial = pooled.serial;
    } else if (cfg->hip_stream)
        ctx->stream = (hipStream_t)cfg->hip_stream;
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(EFX_ERR_DEVICE);
        ctx->own_stream = true;
        ctx->pooled_streams = true;
        std::lock_guard<std::mutex> lk(g_pool_guard);
        ctx->stream_serial = ++g_set_serial;
    }
    if (!ctx->parse_streams[0]) {
        // The parse kernel is a few thousand long-running waves: give it the higher priority so its workgroups are
        // placed as soon as the reconstruction kernels of the previous call free a slot.
        // The runtime creates a stream's hardware queue at its first launch, and queues are dealt onto the
        // command processor's pipes in creation order: left to first use (reconstruction stream, first parse stream,
        // copy stream, and the second parse stream only at the second decode) the second parse stream ends up
        // taking turns with the reconstruction stream instead of running beside it -- 2.2 instead of 1.6 ms per step
        // (measured on the first context of a process, tools/exp/overlap_probe.py).  So every queue is instantiated
        // here, in this order, by an empty launch.
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess)
            return bail(EFX_ERR_DEVICE);
        for (auto& ps : ctx->parse_streams)
            if (hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi) != hipSuccess)
                return bail(EFX_ERR_DEVICE);
        if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
            return bail(EFX_ERR_DEVICE);
        std::vector<hipStream_t> order = {ctx->stream};
        for (auto ps : ctx->parse_streams)
            order.push_back(ps);
        order.push_back(ctx->copy_stream);
        for (hipStream_t st : order) {
            hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, st, (uint32_t*)nullptr, 0u, (size_t)0);
            if (hipStreamSynchronize(st) != hipSuccess)
                return bail(EFX_ERR_DEVICE);
        }
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
    A(dalloc(&ctx->d_tables, 1));
    A(dalloc(&ctx->d_tm_tables, 1));
    A(dalloc(&ctx->d_state, n));
    for (auto& u : ctx->up) {
        A(dalloc(&u.d_es, ctx->es_cap));
        A(dalloc(&u.d_stream_off, n + 1));
        A(dalloc(&u.d_stream_perm, n));
        A(hipHostMalloc(reinterpret_cast<void**>(&u.h_es), ctx->es_cap, hipHostMallocDefault));
        A(hipHostMalloc(reinterpret_cast<void**>(&u.h_meta), meta_bytes(n), hipHostMallocDefault));
        A(hipEventCreateWithFlags(&u.uploaded, hipEventDisableTiming));
        for (auto& ev : u.last_read)
            A(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (e == hipSuccess)
            A(hipMemset(u.d_es, 0, ctx->es_cap));
    }
    for (auto& sl : ctx->slot) {
        A(dalloc(&sl.d_pic_count, n));
        A(dalloc(&sl.d_status, n));
        A(dalloc(&sl.d_counters, kMaxGroups));
        A(dalloc(&sl.d_mbrecs, n * P * kMbCount));
        A(dalloc(&sl.d_raw, n * P * kMbCount));
        A(dalloc(&sl.d_coefs, ctx->es_cap * kCoefsPerEsByte));
        A(dalloc(&sl.d_pics, n * P));
        A(dalloc(&sl.d_slices_tmp, n * P * kMaxSlicesPerPicture));
        A(dalloc(&sl.d_qtab, n * P * 64));
        A(dalloc(&sl.d_slice_base, n * P + kMaxGroups));  // (one list per parse half, each with an end entry)
        A(dalloc(&sl.d_descs, n * P * kMaxSlicesPerPicture));
        A(dalloc(&sl.d_call_pos, 2 * n));
        A(dalloc(&sl.d_recon_sync, kReconSyncWords(n)));
    }
    A(dalloc(&ctx->d_frames, n * D * kFrameBytes + 8192));  // slack: k_recon's window rows may over-read the last frame
    A(dalloc(&ctx->d_video[0], 1));
    A(dalloc(&ctx->d_video[1], 1));
    A(dalloc(&ctx->d_video_lines[0], 1));
    A(dalloc(&ctx->d_video_lines[1], 1));
    A(dalloc(&ctx->d_hash, n * D));
    A(dalloc(&ctx->d_sbc_tables, 1));
    if (e != hipSuccess) {
        fprintf(stderr, "efx_create: %s\n", hipGetErrorString(e));
        return bail(EFX_ERR_DEVICE);
    }
    {
        // parse workgroups resident per parse kernel beside reconstruction launches: five for every eight compute units.
        // Measured on 256 CUs, 1024 streams x GOP 12 (tools/exp/cap_sweep2.py, profiles/r4_schedule_sweep.md): groups of 512
        // streams without a cap 8.05-8.18 M frames/s, capped at 128: 8.15-8.54; groups of 1024 capped at 128 ... 192:
        // 8.57-9.06 (flat between 144 and 192 on one box, best at 176 on another); 96 and below: the parse kernel
        // becomes the critical path.  5-slice pictures (4.4 x the tokens per slice): uncapped 7.21 M, 160: 7.55 M.
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
            ctx->parse_wg_cap = std::max(1, prop.multiProcessorCount * 5 / 8);
            ctx->n_cus = std::max(1, prop.multiProcessorCount);
        }
        if (const char* v = getenv("EFX_GROUPS"))
            ctx->opt_groups = atoi(v);
        if (const char* v = getenv("EFX_FORCE_PARSE_CAP"))
            ctx->opt_parse_cap = atoi(v);
        if (const char* v = getenv("EFX_RECON_MODE"))
            ctx->opt_recon_mode = atoi(v);
        if (const char* v = getenv("EFX_RECON_WAVES"))
            ctx->opt_recon_waves = atoi(v);
        if (const char* v = getenv("EFX_DEMUX_FUSED"))
            ctx->opt_demux_fused = atoi(v) != 0;
        if (const char* v = getenv("EFX_RECON_ITEMS"))
            ctx->opt_recon_items = atoi(v);
        if (const char* cap = getenv("EFX_PARSE_WG_CAP"))  // (development: 0 = no cap)
            ctx->parse_wg_cap = atoi(cap);
    }
    ParseTables* pt = new ParseTables;
    build_parse_tables(pt);
    A(hipMemcpy(ctx->d_tables, pt, sizeof(ParseTables), hipMemcpyHostToDevice));
    {
        TmTables* tt = new TmTables;
        build_tm_tables(tt);
        hipError_t e_tm = hipMemcpy(ctx->d_tm_tables, tt, sizeof(TmTables), hipMemcpyHostToDevice);
        delete tt;
        A(e_tm);
    }
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
    for (auto& sl : ctx->slot) {
        A(hipMemset(sl.d_mbrecs, 0, n * P * kMbCount * sizeof(MbRec)));
        for (auto& ev : sl.parse_done)
            A(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        A(hipEventCreateWithFlags(&sl.recon_done, hipEventDisableTiming));
        A(hipEventCreateWithFlags(&sl.wrap_cleared, hipEventDisableTiming));
    }
    if (e != hipSuccess)
        return bail(EFX_ERR_DEVICE);
    *out = ctx;
    if (efx_reset(ctx) != EFX_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        *out = nullptr;
        return bail(EFX_ERR_DEVICE);
    }
    return EFX_OK;
}

void efx_destroy(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return;
    if (ctx->copy_stream)
        (void)hipStreamSynchronize(ctx->copy_stream);
    for (auto ps : ctx->parse_streams)
        if (ps)
            (void)hipStreamSynchronize(ps);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (auto& a : ctx->arenas) {
        if (a.owned)
            (void)hipHostFree(a.base);
        else
            (void)hipHostUnregister(a.base);
    }
    void* bufs[] = {ctx->d_tables, ctx->d_tm_tables, ctx->d_sbc_flags, ctx->d_sbc_next, ctx->d_sbc_info, ctx->d_sbc_plan, ctx->d_sbc_extra, ctx->d_sbc_cover, ctx->d_state, ctx->d_frames, ctx->d_video[0],  ctx->d_video[1], ctx->d_video_lines[0],
                    ctx->d_video_lines[1], ctx->d_hash, ctx->d_ts, ctx->d_demux_chunks, ctx->d_sbc_tables, ctx->d_idx_info, ctx->d_ts_off, ctx->d_idx_len,
                    ctx->d_idx_base, ctx->d_idx_seq};
    for (auto& te : ctx->timing_ring)
        for (auto& ev : te.ev)
            if (ev)
                (void)hipEventDestroy(ev);
    for (void* b : bufs)
        if (b)
            (void)dev_free(b);
    for (auto& u : ctx->up) {
        void* ub[] = {u.d_es, u.d_stream_off, u.d_stream_perm, u.d_ts_len, u.d_pkt_base, u.d_es_len, u.d_pes_count, u.d_pes};
        for (void* b : ub)
            if (b)
                (void)dev_free(b);
        if (u.h_es)
            (void)hipHostFree(u.h_es);
        if (u.h_meta)
            (void)hipHostFree(u.h_meta);
        hipEvent_t evs[] = {u.uploaded, u.ev_demux[0], u.ev_demux[1]};
        for (auto ev : evs)
            if (ev)
                (void)hipEventDestroy(ev);
        for (auto ev : u.last_read)
            if (ev)
                (void)hipEventDestroy(ev);
    }
    for (auto& sl : ctx->slot) {
        void* sb[] = {sl.d_pic_count, sl.d_status, sl.d_counters, sl.d_mbrecs, sl.d_raw, sl.d_coefs, sl.d_pts, sl.d_call_pos, sl.d_recon_sync,
                      sl.d_pics,      sl.d_slices_tmp, sl.d_qtab, sl.d_slice_base, sl.d_descs};
        for (void* b : sb)
            if (b)
                (void)dev_free(b);
        for (auto ev : sl.parse_done)
            if (ev)
                (void)hipEventDestroy(ev);
        if (sl.recon_done)
            (void)hipEventDestroy(sl.recon_done);
        if (sl.wrap_cleared)
            (void)hipEventDestroy(sl.wrap_cleared);
    }
    bool complete = ctx->own_stream && ctx->stream && ctx->copy_stream;
    for (auto ps : ctx->parse_streams)
        complete = complete && ps;
    if (ctx->pooled_streams && complete) {  // (all of them idle: synchronised above)
        StreamSet set;
        set.recon = ctx->stream;
        for (int i = 0; i < kParseStreams; i++)
            set.parse[i] = ctx->parse_streams[i];
        set.copy = ctx->copy_stream;
        set.serial = ctx->stream_serial;
        park_stream_set(ctx->cfg.device, set);
    } else {
        for (auto ps : ctx->parse_streams)
            if (ps)
                (void)hipStreamDestroy(ps);
        if (ctx->copy_stream)
            (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->own_stream && ctx->stream)
            (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}


// transport-stream buffers (first EFX_FORMAT_TS upload or first index call): the TS of stream i
// occupies the same region of d_ts that its elementary stream will occupy in d_es (an ES is never
// longer than its TS)
static int ensure_ts_buffers(efx_ctx* ctx)
{
    if (ctx->ts_ready)
        return EFX_OK;
    if (ctx->d_ts)
        return fail(ctx, EFX_ERR_DEVICE, "transport-stream buffers: an earlier allocation failed");
    const size_t n_max = (size_t)ctx->cfg.max_streams;
    ctx->pes_cap = ctx->es_cap / 188 + n_max;
    hipError_t e = dalloc(&ctx->d_ts, ctx->es_cap);
    if (e == hipSuccess)
        // (the chunk descriptors, and behind them one ticket counter per stream for the one-pass kernel: zeroed together)
        e = dev_alloc(reinterpret_cast<void**>(&ctx->d_demux_chunks), (ctx->pes_cap / kDemuxChunk + n_max + 2) * 16 + n_max * sizeof(uint32_t));
    for (auto& u : ctx->up) {
        if (e == hipSuccess) e = dalloc(&u.d_ts_len, n_max);
        if (e == hipSuccess) e = dalloc(&u.d_pkt_base, n_max);
        if (e == hipSuccess) e = dalloc(&u.d_es_len, n_max);
        if (e == hipSuccess) e = dalloc(&u.d_pes_count, n_max);
        if (e == hipSuccess) e = dalloc(&u.d_pes, ctx->pes_cap);
        for (auto& ev : u.ev_demux)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_info, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_ts_off, n_max + 1);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_len, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_base, n_max);
    if (e == hipSuccess) e = dalloc(&ctx->d_idx_seq, ctx->pes_cap);
    for (auto& sl : ctx->slot)
        if (e == hipSuccess) e = dalloc(&sl.d_pts, n_max * (size_t)ctx->cfg.max_pictures + n_max);
    if (e != hipSuccess)
        return fail(ctx, EFX_ERR_DEVICE, "transport-stream buffers", e);
    ctx->ts_ready = true;
    return EFX_OK;
}

// efx_upload_streams (in_place = false: the caller's bytes are only ever READ, whatever the pointers look like) and
// efx_upload_streams_inplace (the batch lies in an arena in the device layout; the library writes the tails and the zero fill
// into the gaps of the layout and the transfer reads the arena)
static int upload_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* data, const size_t* len, int format, bool in_place)
{
    bind_device(ctx);
    if (!ctx || !data || !len || n_streams <= 0 || (format != EFX_FORMAT_ES && format != EFX_FORMAT_TS))
        return fail(ctx, EFX_ERR_ARG, "efx_upload_streams: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: more streams than max_streams");
    const bool is_ts = format == EFX_FORMAT_TS;
    if (is_ts) {
        int r = ensure_ts_buffers(ctx);
        if (r)
            return r;
    }
    static const uint8_t tail[kEsTailBytes] = {0, 0, 0, 1, 0xB7, 0, 0, 1, 0xB7};
    // offsets and lengths are built in locals and moved into the batch only when the call can no longer be
    // rejected: a refused upload leaves the resident batch (and what efx_download_es reads) untouched
    std::vector<uint64_t> stream_off((size_t)n_streams + 1, 0);
    std::vector<uint32_t> es_len((size_t)n_streams, 0);
    size_t pos = 0, packets = 0;
    for (int i = 0; i < n_streams; i++) {
        const size_t n = len[i];
        if (!data[i] && n)
            return fail(ctx, EFX_ERR_ARG, "efx_upload_streams: null stream");
        const size_t padded = (n + kEsTailBytes + 15) & ~(size_t)15;
        if (pos + padded + kEsGuardBytes > ctx->es_cap)
            return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: more bytes than max_stream_bytes");
        stream_off[i] = pos;
        es_len[i] = (uint32_t)n;
        packets += n / 188;
        pos += padded;
    }
    stream_off[n_streams] = pos;
    if (is_ts && packets > ctx->pes_cap)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_upload_streams: PES list capacity");

    // the buffer the last decode does not read; its previous content may still be in use by an older parse
    // half (two calls back) and its staging memory by its own H2D transfer
    const int ui = (ctx->cur_up + 1) % kUploads;
    efx_ctx::Upload& u = ctx->up[ui];
    hipStream_t st = ctx->copy_stream;
    // in place (asked for by name, round-5 ADVICE: never inferred from what the pointers look like -- a caller that packs
    // streams back to back in page-locked memory and uploads them one at a time would have had the next stream's first bytes
    // overwritten by this stream's tail): the batch must lie in one arena exactly as the device buffer holds it
    // (efx_stream_layout), else the call is refused
    const uint8_t* arena_src = nullptr;
    if (in_place) {
        for (const auto& a : ctx->arenas)
            if (data[0] >= a.base && data[0] + pos <= a.base + a.bytes) {
                arena_src = data[0];
                for (int i = 1; i < n_streams && arena_src; i++)
                    if (data[i] != data[0] + stream_off[i])
                        arena_src = nullptr;
                break;
            }
        if (!arena_src)
            return fail(ctx, EFX_ERR_ARG, "efx_upload_streams_inplace: the batch does not lie in an arena of this context in the layout of efx_stream_layout");
    }
    if (u.valid) {
        // this record's pinned metadata (and, staged path, its staging buffer) are about to be rewritten by the host: its own
        // previous transfer must have read them -- three uploads ago, long done
        EFX_HIP(hipEventSynchronize(u.uploaded));
        // ... and the device buffer is free once the parse halves that read its previous batch are through (three uploads ago:
        // long done; letting the copy stream wait instead of the host was measured and is slower, 6.9 against 8.0 M frames/s)
        for (auto ev : u.last_read)
            EFX_HIP(hipEventSynchronize(ev));
    }
    u.valid = false;
    uint8_t* d_dst = is_ts ? ctx->d_ts : u.d_es;
    // small per-stream arrays, pinned: stream_off | perm | ts_len | pkt_base
    uint64_t* m_off = reinterpret_cast<uint64_t*>(u.h_meta);
    uint32_t* m_perm = reinterpret_cast<uint32_t*>(m_off + n_streams + 1);
    uint32_t *m_ts_len = m_perm + n_streams, *m_pkt_base = m_ts_len + n_streams;
    memcpy(m_off, stream_off.data(), ((size_t)n_streams + 1) * sizeof(uint64_t));
    {
        size_t pk = 0;
        for (int i = 0; i < n_streams; i++) {
            m_ts_len[i] = (uint32_t)len[i];
            m_pkt_base[i] = (uint32_t)pk;
            pk += len[i] / 188;
        }
    }
    // Stage into the pinned buffer and ship it: the streams are cut into groups, one host thread
    // copies each group, and a group's H2D transfer is queued as soon as its copy is done, so the
    // transfer of the first groups runs under the copies of the last.
    auto stage = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            uint8_t* at = u.h_es + stream_off[i];
            const size_t n = len[i], padded = (size_t)(stream_off[i + 1] - stream_off[i]);
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
    if (arena_src) {
        // the gaps of the layout are the library's: end-of-data tail (ES) and zero fill; then ONE transfer from caller memory
        uint8_t* w = const_cast<uint8_t*>(arena_src);
        for (int i = 0; i < n_streams; i++) {
            uint8_t* at = w + stream_off[i];
            const size_t n = len[i], padded = (size_t)(stream_off[i + 1] - stream_off[i]);
            if (is_ts)
                memset(at + n, 0, padded - n);
            else {
                memcpy(at + n, tail, kEsTailBytes);
                memset(at + n + kEsTailBytes, 0, padded - n - kEsTailBytes);
            }
        }
        if (pos)
            EFX_HIP(hipMemcpyAsync(d_dst, arena_src, pos, hipMemcpyHostToDevice, st));
    } else {
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
            const size_t a = stream_off[first[g]], b = stream_off[first[g + 1]];
            if (e == hipSuccess && b > a)
                e = hipMemcpyAsync(d_dst + a, u.h_es + a, b - a, hipMemcpyHostToDevice, st);
        }
        if (e != hipSuccess)
            return fail(ctx, EFX_ERR_DEVICE, "efx_upload_streams: H2D", e);
    }
    // slices of one picture index are dealt to the parse waves stream by stream: longest streams
    // first, so that a wave's 64 slices have similar bit rates (and the long waves start early)
    for (int i = 0; i < n_streams; i++)
        m_perm[i] = (uint32_t)i;
    {
        // (sorted inside every group of streams that efx_decode runs as a unit: the groups of an idle-GPU call, or the
        // pinned number of them; a call that runs as fewer, coarser groups is served by the same order)
        const int G = ctx->opt_groups > 0 ? std::min(std::min(ctx->opt_groups, kMaxGroups), std::max(1, n_streams / 8)) : group_count(n_streams);
        u.sort_groups = G;
        for (int g = 0; g < G; g++)
            std::stable_sort(m_perm + group_first(n_streams, G, g), m_perm + group_first(n_streams, G, g + 1),
                             [&](uint32_t a, uint32_t b) { return len[a] > len[b]; });
    }
    EFX_HIP(hipMemcpyAsync(u.d_stream_off, m_off, ((size_t)n_streams + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    EFX_HIP(hipMemcpyAsync(u.d_stream_perm, m_perm, (size_t)n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    u.ts_bytes = 0;
    u.demux_timed = false;
    if (is_ts) {
        EFX_HIP(hipMemsetAsync(ctx->d_ts + pos, 0, kEsGuardBytes, st));
        EFX_HIP(hipMemcpyAsync(u.d_ts_len, m_ts_len, n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        EFX_HIP(hipMemcpyAsync(u.d_pkt_base, m_pkt_base, n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        EFX_HIP(hipMemsetAsync(u.d_es + pos, 0, kEsGuardBytes, st));
        // MpegDecoder::more()/demux() for the whole batch (player.cpp:381-493)
        if (ctx->timing)
            EFX_HIP(hipEventRecord(u.ev_demux[0], st));
        {
            // chunk-parallel: per-chunk totals, the chain across a stream's chunks (and the end-of-data tail), the gather
            size_t max_len = 0;
            for (int i = 0; i < n_streams; i++)
                max_len = std::max(max_len, len[i]);
            const unsigned chunks = (unsigned)std::max<size_t>(1, (max_len / 188 + kDemuxChunk - 1) / kDemuxChunk);
            if (kDemuxThreads == 64 && ctx->opt_demux_fused) {
                // one pass, one launch: tickets + decoupled look-back over the chunk descriptors (k_demux.hip); both zeroed here.
                // The tickets sit in front of the descriptors this batch uses, so that ONE memset covers them.
                const size_t n_desc = packets / kDemuxChunk + (size_t)n_streams + 2;
                uint8_t* const desc_bytes = reinterpret_cast<uint8_t*>(ctx->d_demux_chunks);
                uint32_t* tickets = reinterpret_cast<uint32_t*>(desc_bytes + (ctx->pes_cap / kDemuxChunk + (size_t)ctx->cfg.max_streams + 2) * 16);
                EFX_HIP(hipMemsetAsync(desc_bytes, 0, n_desc * 16, st));
                EFX_HIP(hipMemsetAsync(tickets, 0, (size_t)n_streams * sizeof(uint32_t), st));
                hipLaunchKernelGGL(k_demux_fused, dim3(chunks, n_streams), dim3(kDemuxThreads), 0, st, ctx->d_ts, u.d_stream_off, u.d_ts_len,
                                   u.d_pkt_base, ctx->d_demux_chunks, u.d_es, u.d_pes, tickets, u.d_es_len, u.d_pes_count);
            } else {
                hipLaunchKernelGGL(k_demux_scan, dim3(chunks, n_streams), dim3(kDemuxThreads), 0, st, ctx->d_ts, u.d_stream_off, u.d_ts_len, u.d_pkt_base,
                                   ctx->d_demux_chunks);
                hipLaunchKernelGGL(k_demux_offsets, dim3(n_streams), dim3(64), 0, st, u.d_ts_len, u.d_pkt_base, ctx->d_demux_chunks, u.d_es,
                                   u.d_stream_off, u.d_es_len, u.d_pes_count);
                hipLaunchKernelGGL(k_demux, dim3(chunks, n_streams), dim3(kDemuxThreads), 0, st, ctx->d_ts, u.d_stream_off, u.d_ts_len, u.d_pkt_base,
                                   ctx->d_demux_chunks, u.d_es, u.d_pes);
            }
        }
        if (ctx->timing) {
            EFX_HIP(hipEventRecord(u.ev_demux[1], st));
            u.demux_timed = true;
        }
        EFX_HIP(hipGetLastError());
        for (int i = 0; i < n_streams; i++)
            u.ts_bytes += len[i];
    } else
        EFX_HIP(hipMemsetAsync(u.d_es + pos, 0, kEsGuardBytes, st));
    EFX_HIP(hipEventRecord(u.uploaded, st));
    // the batch is resident (as far as every later call on this context is concerned: they wait for the event)
    u.stream_off.swap(stream_off);
    u.es_len.swap(es_len);
    u.n_streams = n_streams;
    u.es_used = pos;
    u.ts_input = is_ts;
    u.valid = true;
    ctx->cur_up = ui;
    return EFX_OK;
}

int efx_upload_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* data, const size_t* len, int format)
{
    return upload_streams(ctx, n_streams, data, len, format, false);
}

int efx_upload_streams_inplace(efx_ctx* ctx, int n_streams, uint8_t* const* data, const size_t* len, int format)
{
    return upload_streams(ctx, n_streams, data, len, format, true);
}


int efx_stream_layout(int n_streams, const size_t* len, size_t* offsets)
{
    if (n_streams <= 0 || !len || !offsets)
        return EFX_ERR_ARG;
    size_t pos = 0;
    for (int i = 0; i < n_streams; i++) {
        offsets[i] = pos;
        pos += (len[i] + kEsTailBytes + 15) & ~(size_t)15;  // (the rule efx_upload_streams places streams by)
    }
    offsets[n_streams] = pos;
    return EFX_OK;
}

int efx_host_alloc(efx_ctx* ctx, size_t bytes, void** host_ptr)
{
    bind_device(ctx);
    if (!ctx || !host_ptr || !bytes)
        return EFX_ERR_ARG;
    void* p = nullptr;
    EFX_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    ctx->arenas.push_back({static_cast<uint8_t*>(p), bytes, true});
    *host_ptr = p;
    return EFX_OK;
}

int efx_host_register(efx_ctx* ctx, void* host_ptr, size_t bytes)
{
    bind_device(ctx);
    if (!ctx || !host_ptr || !bytes)
        return EFX_ERR_ARG;
    EFX_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    ctx->arenas.push_back({static_cast<uint8_t*>(host_ptr), bytes, false});
    return EFX_OK;
}

static int drop_arena(efx_ctx* ctx, void* host_ptr, bool owned)
{
    bind_device(ctx);
    if (!ctx || !host_ptr)
        return EFX_ERR_ARG;
    for (size_t i = 0; i < ctx->arenas.size(); i++)
        if (ctx->arenas[i].base == host_ptr && ctx->arenas[i].owned == owned) {
            // a transfer may still be reading it
            if (ctx->copy_stream)
                EFX_HIP(hipStreamSynchronize(ctx->copy_stream));
            ctx->arenas.erase(ctx->arenas.begin() + (ptrdiff_t)i);
            if (owned)
                EFX_HIP(hipHostFree(host_ptr));
            else
                EFX_HIP(hipHostUnregister(host_ptr));
            return EFX_OK;
        }
    return fail(ctx, EFX_ERR_ARG, owned ? "efx_host_free: not an arena of this context" : "efx_host_unregister: not registered with this context");
}

int efx_host_free(efx_ctx* ctx, void* host_ptr) { return drop_arena(ctx, host_ptr, true); }
int efx_host_unregister(efx_ctx* ctx, void* host_ptr) { return drop_arena(ctx, host_ptr, false); }

int efx_upload_done(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    if (ctx->cur_up < 0)
        return 1;
    hipError_t e = hipEventQuery(ctx->up[ctx->cur_up].uploaded);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    return e == hipSuccess ? 1 : fail(ctx, EFX_ERR_DEVICE, "efx_upload_done", e);
}

int efx_set_option(efx_ctx* ctx, int option, int value)
{
    if (!ctx || value < 0)
        return EFX_ERR_ARG;
    switch (option) {
    case EFX_OPT_GROUPS:
        if (value > kMaxGroups)
            return EFX_ERR_ARG;
        ctx->opt_groups = value;
        return EFX_OK;
    case EFX_OPT_PARSE_CAP:
        if (value > 2)
            return EFX_ERR_ARG;
        ctx->opt_parse_cap = value;
        return EFX_OK;
    case EFX_OPT_RECON_MODE:
        if (value > 2)
            return EFX_ERR_ARG;
        ctx->opt_recon_mode = value;
        return EFX_OK;
    case EFX_OPT_RECON_WAVES:
        if (value > 64)
            return EFX_ERR_ARG;
        ctx->opt_recon_waves = value;
        return EFX_OK;
    case EFX_OPT_RECON_ITEMS:
        ctx->opt_recon_items = value;
        return EFX_OK;
    case EFX_OPT_SBC_SERIAL:
        ctx->opt_sbc_serial = value != 0;
        return EFX_OK;
    case EFX_OPT_DEMUX_FUSED:
        ctx->opt_demux_fused = value != 0;
        return EFX_OK;
    default:
        return EFX_ERR_ARG;
    }
}

int efx_get_option(efx_ctx* ctx, int option, int* value)
{
    bind_device(ctx);
    if (!ctx || !value)
        return EFX_ERR_ARG;
    switch (option) {
    case EFX_OPT_GROUPS: *value = ctx->opt_groups; return EFX_OK;
    case EFX_OPT_PARSE_CAP: *value = ctx->opt_parse_cap; return EFX_OK;
    case EFX_OPT_RECON_MODE: *value = ctx->opt_recon_mode; return EFX_OK;
    case EFX_OPT_RECON_WAVES: *value = ctx->opt_recon_waves; return EFX_OK;
    case EFX_OPT_RECON_ITEMS: *value = ctx->opt_recon_items; return EFX_OK;
    case EFX_OPT_SBC_SERIAL: *value = ctx->opt_sbc_serial; return EFX_OK;
    case EFX_OPT_DEMUX_FUSED: *value = ctx->opt_demux_fused; return EFX_OK;
    case EFX_OPT_RECON_SPINS: {
        int r = sync_all(ctx);
        if (r)
            return r;
        uint32_t total = 0;
        for (int g = 0; g < ctx->n_groups && ctx->decoded; g++) {
            uint32_t v = 0;
            EFX_HIP(hipMemcpy(&v, ctx->slot[ctx->groups[g].slot].d_recon_sync + 8 * 32, sizeof(v), hipMemcpyDeviceToHost));
            total += v;
        }
        *value = (int)std::min<uint32_t>(total, 0x7FFFFFFFu);
        return EFX_OK;
    }
    default:
        return EFX_ERR_ARG;
    }
}

int efx_download_es(efx_ctx* ctx, int stream, uint8_t* dst, size_t cap, size_t* es_len)
{
    bind_device(ctx);
    if (!ctx || !es_len || ctx->cur_up < 0 || stream < 0 || stream >= ctx->up[ctx->cur_up].n_streams || (!dst && cap))
        return ctx && ctx->cur_up < 0 ? fail(ctx, EFX_ERR_STATE, "efx_download_es: no streams uploaded") : EFX_ERR_ARG;
    efx_ctx::Upload& u = ctx->up[ctx->cur_up];
    EFX_HIP(hipEventSynchronize(u.uploaded));
    size_t n;
    if (u.ts_input) {
        uint32_t v = 0;
        EFX_HIP(hipMemcpy(&v, u.d_es_len + stream, sizeof(v), hipMemcpyDeviceToHost));
        n = v;
    } else
        n = u.es_len[stream];
    *es_len = n;
    size_t c = n < cap ? n : cap;
    if (c)
        EFX_HIP(hipMemcpy(dst, u.d_es + u.stream_off[stream], c, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_reset(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    const size_t n = (size_t)ctx->cfg.max_streams;
    EFX_HIP(hipMemsetAsync(ctx->d_frames, 0, n * ctx->cfg.ring_depth * kFrameBytes, ctx->stream));
    // the constructor's state (player.cpp:354-361): frame index 1 (_reference = _fb[0], _current = _fb[1]), no PTS seen
    std::vector<StreamState> init(n);
    for (auto& st : init) {
        st.fb_index = 1;
        st.pts_seen = 0;
        st.pts_carry = -1;
    }
    EFX_HIP(hipMemcpyAsync(ctx->d_state, init.data(), n * sizeof(StreamState), hipMemcpyHostToDevice, ctx->stream));
    EFX_HIP(hipStreamSynchronize(ctx->stream));  // (pageable source)
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    return EFX_OK;
}

int efx_play_reset(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    int r = sync_all(ctx);
    if (r)
        return r;
    const size_t n = (size_t)ctx->cfg.max_streams;
    std::vector<StreamState> st(n);
    EFX_HIP(hipMemcpy(st.data(), ctx->d_state, n * sizeof(StreamState), hipMemcpyDeviceToHost));
    for (auto& x : st)
        x.pts_seen = 0;  // _last_pts = -1; _fb_index and _pts survive (player.cpp:439-453)
    EFX_HIP(hipMemcpy(ctx->d_state, st.data(), n * sizeof(StreamState), hipMemcpyHostToDevice));
    return EFX_OK;
}

int efx_stream_state(efx_ctx* ctx, int stream, uint32_t* frame_index, int* pts_seen, int64_t* newest_pts)
{
    bind_device(ctx);
    if (!ctx || stream < 0 || stream >= ctx->cfg.max_streams)
        return EFX_ERR_ARG;
    int r = sync_all(ctx);
    if (r)
        return r;
    StreamState st;
    EFX_HIP(hipMemcpy(&st, ctx->d_state + stream, sizeof(st), hipMemcpyDeviceToHost));
    if (frame_index)
        *frame_index = st.fb_index;
    if (pts_seen)
        *pts_seen = (int)st.pts_seen;
    if (newest_pts)
        *newest_pts = st.pts_carry;
    return EFX_OK;
}

int efx_erase_frames(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    size_t bytes = (size_t)ctx->cfg.max_streams * ctx->cfg.ring_depth * kFrameBytes;
    EFX_HIP(hipMemsetAsync(ctx->d_frames, 0x30, bytes, ctx->stream));
    return EFX_OK;
}

int efx_decode_from(efx_ctx* ctx, int first_picture) { return efx_decode_range(ctx, first_picture, ctx ? ctx->cfg.max_pictures : 0); }

int efx_decode_range(efx_ctx* ctx, int first_picture, int n_pictures)
{
    bind_device(ctx);
    if (!ctx || first_picture < 0 || n_pictures < 1 || n_pictures > ctx->cfg.max_pictures)
        return EFX_ERR_ARG;
    if (ctx->cur_up < 0)
        return fail(ctx, EFX_ERR_STATE, "efx_decode: no streams uploaded");
    efx_ctx::Upload& u = ctx->up[ctx->cur_up];
    const int n_all = u.n_streams, P = ctx->cfg.max_pictures, D = ctx->cfg.ring_depth;
    // Back-to-back calls pipeline against each other -- the parse half of call n + 1 (and n + 2) runs beside the
    // reconstruction of call n -- and then ONE group per call is best whatever the batch (4096 streams, 12-slice pictures:
    // 9.3 M frames/s as one group, 9.15 M as four; 5-slice pictures 7.85 / 7.57 M; profiles/r4_schedule_sweep.md): fewer,
    // fatter reconstruction launches have fewer draining tails.  A call that finds the reconstruction stream idle (one call
    // at a time) has nothing to pipeline against but itself: it runs as groups of kGroupStreams streams, the parse half of
    // group g + 1 beside the reconstruction of group g.  Results do not depend on the split.
    const bool idle_at_call = ctx->recon_groups_unsynced == 0;  // (by the library's own count: efx_ctx::recon_groups_unsynced)
    // (efx_set_option pins the structure: the same call sequence then runs the same launches whatever the timing)
    const int G = ctx->opt_groups > 0 ? std::min(std::min(ctx->opt_groups, kMaxGroups), std::max(1, n_all / 8))
                                      : (idle_at_call ? group_count(n_all) : 1);
    {
        // The order of the slices inside a picture index (streams by descending length) was laid down at upload INSIDE the
        // groups of that moment; every group of this call must be a union of those -- else (EFX_OPT_GROUPS changed between
        // upload and decode) the order is laid down again for this call's groups, once, behind everything queued.
        bool nested = true;
        for (int g = 1; g < G && nested; g++) {
            bool found = false;
            for (int k = 0; k <= u.sort_groups && !found; k++)
                found = group_first(n_all, u.sort_groups, k) == group_first(n_all, G, g);
            nested = found;
        }
        if (!nested) {
            int r = sync_all(ctx);
            if (r)
                return r;
            std::vector<uint32_t> perm((size_t)n_all);
            for (int i = 0; i < n_all; i++)
                perm[i] = (uint32_t)i;
            auto len_of = [&](uint32_t i) { return u.stream_off[i + 1] - u.stream_off[i]; };  // (padded lengths: the same order up to ties)
            for (int g = 0; g < G; g++)
                std::stable_sort(perm.begin() + group_first(n_all, G, g), perm.begin() + group_first(n_all, G, g + 1),
                                 [&](uint32_t a, uint32_t b) { return len_of(a) > len_of(b); });
            EFX_HIP(hipMemcpy(u.d_stream_perm, perm.data(), perm.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            u.sort_groups = G;
        }
    }
    hipStream_t sr = ctx->stream;
    int timing_slot = -1;
    if (ctx->timing && !ctx->timing_ring.empty()) {
        timing_slot = (int)(ctx->timed_calls++ % kTimingRing);
        ctx->timing_groups[timing_slot] = (uint8_t)G;
        ctx->timing_leaders[timing_slot] = 0;
        ctx->timing_launches[timing_slot] = (uint16_t)(((G + kReconMerge - 1) / kReconMerge) * (ctx->opt_recon_mode == 0 ? n_pictures : 1));
    }
    ctx->last_upload = ctx->cur_up;
    ctx->last_n_streams = u.n_streams;
    ctx->last_ts_input = u.ts_input;
    // G parse halves; kReconMerge consecutive halves form one reconstruction group: they parse side by side on the parse
    // streams into disjoint stream ranges of ONE hand-over slot, and one k_recon launch per picture index covers them all
    // (a launch is a barrier: the fewer and fatter the launches, the smaller the share of their draining tails)
    const int R = (G + kReconMerge - 1) / kReconMerge;
    ctx->n_groups = R;
    for (int r = 0; r < R; r++) {
        const int h0 = r * kReconMerge, h1 = std::min(G, h0 + kReconMerge);
        const int rs0 = group_first(n_all, G, h0), rn = group_first(n_all, G, h1) - rs0;
        const int slot = (int)(ctx->subcalls++ % kSlots);
        ctx->groups[r] = {slot, rs0, rn, h0, h1 - h0};
        efx_ctx::Slot& sl = ctx->slot[slot];
        sl.upload = ctx->cur_up;
        // macroblock records carry the epoch that wrote them; recycle the tag space by clearing
        const bool wrap = ++sl.epoch > 255;
        if (wrap)
            sl.epoch = 1;
        efx_ctx::TimingEvents* te0 = nullptr;
        for (int g = h0; g < h1; g++) {
            const int s0 = group_first(n_all, G, g), n = group_first(n_all, G, g + 1) - s0;
            const int pi = (int)(ctx->parse_calls++ % kParseStreams);
            hipStream_t sp = ctx->parse_streams[pi];
            DecodeCounters* counters = sl.d_counters + g;
            uint32_t* slice_base = sl.d_slice_base + (size_t)s0 * P + g;               // this half's own lists
            SliceDesc* descs = sl.d_descs + (size_t)s0 * P * kMaxSlicesPerPicture;

            // ---- parse half (a parse stream): index -> slice list -> VLC parse -----------------------------------
            EFX_HIP(hipStreamWaitEvent(sp, u.uploaded, 0));     // the batch is in HBM (H2D and k_demux on the copy stream)
            EFX_HIP(hipStreamWaitEvent(sp, sl.recon_done, 0));  // the group kSlots back has released this slot
            // (the WHOLE slot: its stream ranges belong to other halves in other calls, and a record the current parse
            // does not write must never match a tag of the previous 255-cycle)
            if (wrap && g == h0) {
                EFX_HIP(hipMemsetAsync(sl.d_mbrecs, 0, (size_t)ctx->cfg.max_streams * P * kMbCount * sizeof(MbRec), sp));
                EFX_HIP(hipEventRecord(sl.wrap_cleared, sp));
            } else if (wrap)
                EFX_HIP(hipStreamWaitEvent(sp, sl.wrap_cleared, 0));
            efx_ctx::TimingEvents* te = timing_slot >= 0 ? &ctx->timing_ring[(size_t)timing_slot * kMaxGroups + g] : nullptr;
            if (te && !te->ev[0])
                for (auto& ev : te->ev)
                    EFX_HIP(hipEventCreate(&ev));
            if (g == h0)
                te0 = te;
            if (te)
                EFX_HIP(hipEventRecord(te->ev[0], sp));
            hipLaunchKernelGGL(k_index, dim3(n), dim3(64 * kIndexWaves), 0, sp, u.d_es, u.d_stream_off, P, sl.d_pics, sl.d_slices_tmp,
                               sl.d_pic_count, sl.d_status, sl.d_qtab, ctx->d_tables->scan, u.d_pes, u.d_pkt_base, u.d_pes_count,
                               u.ts_input ? sl.d_pts : nullptr, u.ts_input ? sl.d_pts + (size_t)ctx->cfg.max_streams * P : nullptr,
                               first_picture, s0, n_pictures);
            hipLaunchKernelGGL(k_slice_scan, dim3(1), dim3(1024), 0, sp, sl.d_pics, sl.d_pic_count, n, P, u.d_stream_perm + s0,
                               slice_base, counters);
            hipLaunchKernelGGL(k_slice_emit, dim3((n * P * kMaxSlicesPerPicture + 255) / 256), dim3(256), 0, sp, sl.d_pics,
                               sl.d_slices_tmp, sl.d_pic_count, u.d_stream_off, slice_base, n, P, u.d_stream_perm + s0, descs, sl.d_status);
            if (te)
                EFX_HIP(hipEventRecord(te->ev[1], sp));
            const int max_slices = n * P * kMaxSlicesPerPicture;
            const int parse_waves = (max_slices + kParseLanes - 1) / kParseLanes;  // (rounded UP: one lane per slice slot)
            // k_parse's waves pull groups of slices off a counter: the grid is how many of them are resident at a time (above)
            int parse_wgs = (parse_waves + kParseWaves - 1) / kParseWaves;
            // The cap protects the reconstruction launches the parser runs beside.  When the reconstruction stream has
            // nothing queued at this moment -- one call at a time, the first call after a synchronisation -- the parser
            // may have the chip: 0.84 instead of 1.38 ms per 1024 streams x 12 pictures.
            const bool recon_busy = ctx->opt_parse_cap == 1 || (ctx->opt_parse_cap == 0 && ctx->recon_groups_unsynced > 0);
            if (ctx->parse_wg_cap > 0 && recon_busy)
                parse_wgs = std::min(parse_wgs, ctx->parse_wg_cap);
            hipLaunchKernelGGL(k_parse, dim3(parse_wgs), dim3(64 * kParseWaves), 0, sp, u.d_es,
                               descs, counters, ctx->d_tm_tables, sl.d_mbrecs, sl.d_raw, sl.d_coefs, sl.d_status, P, sl.epoch);
            if (te)
                EFX_HIP(hipEventRecord(te->ev[2], sp));
            EFX_HIP(hipEventRecord(sl.parse_done[g - h0], sp));
            EFX_HIP(hipEventRecord(u.last_read[pi], sp));  // the bitstream buffer is free for the upload after next
        }

        // ---- reconstruction of the group (context stream): one launch per picture index --------------------------
        for (int g = h0; g < h1; g++)
            EFX_HIP(hipStreamWaitEvent(sr, sl.parse_done[g - h0], 0));
        if (te0) {
            EFX_HIP(hipEventRecord(te0->ev[4], sr));
            ctx->timing_leaders[timing_slot] |= (uint16_t)(1u << h0);
        }
        // ring positions of this call's pictures; the reconstruction stream orders the calls
        hipLaunchKernelGGL(k_advance, dim3((rn + 255) / 256), dim3(256), 0, sr, ctx->d_state, sl.d_pic_count,
                           u.ts_input ? sl.d_pts : nullptr, u.ts_input ? sl.d_pts + (size_t)ctx->cfg.max_streams * P : nullptr, rs0, rn,
                           P, sl.d_call_pos);
        constexpr int kGroupsPerPicture = (kMbCount * 6 + 63) / 64;
#ifdef EFX_RECON_ITEMS2
        constexpr int kReconGridY = (kGroupsPerPicture + 1) / 2;  // (experiment: two block groups per wave, k_recon.hip)
#else
        constexpr int kReconGridY = kGroupsPerPicture;
#endif
        if (ctx->opt_recon_mode == 0) {
            for (int p = 0; p < n_pictures; p++)
                hipLaunchKernelGGL(k_recon, dim3(rn, kReconGridY), dim3(64), 0, sr, sl.d_mbrecs, sl.d_coefs,
                                   ctx->d_tables->scan, sl.d_qtab, ctx->d_frames, P, D, p, sl.d_call_pos, sl.epoch, rs0);
        } else if (rn > 0) {
            // ONE launch for all picture indices of the group: persistent waves pull (picture, stream, block group) items in
            // picture-major order; a per-stream counter of finished items orders a picture behind its predecessor (k_recon.hip)
            EFX_HIP(hipMemsetAsync(sl.d_recon_sync, 0, kReconSyncWords((size_t)rn) * sizeof(uint32_t), sr));
            // A wave takes opt_recon_items items and ends, so that wave slots (and their LDS) keep coming free for the parse
            // kernel of the next call; with 0 the grid is what the chip holds and the waves live until the last item.
            const long long items = (long long)rn * kGroupsPerPicture * n_pictures;
            const int per_cu = ctx->opt_recon_waves > 0 ? ctx->opt_recon_waves : 18;
            const long long waves = ctx->opt_recon_items > 0 ? (items + ctx->opt_recon_items - 1) / ctx->opt_recon_items
                                                             : std::min<long long>(items, (long long)ctx->n_cus * per_cu);
            hipLaunchKernelGGL(k_recon_all, dim3((unsigned)waves), dim3(64), 0, sr, sl.d_mbrecs,
                               sl.d_coefs, ctx->d_tables->scan, sl.d_qtab, ctx->d_frames, P, D, n_pictures, sl.d_call_pos, sl.epoch, rs0, rn,
                               sl.d_recon_sync, sl.d_status, ctx->opt_recon_items);
        }
        if (te0)
            EFX_HIP(hipEventRecord(te0->ev[3], sr));
        EFX_HIP(hipEventRecord(sl.recon_done, sr));
        ctx->last_recon_done = sl.recon_done;
        ctx->recon_groups_unsynced++;
    }
    EFX_HIP(hipGetLastError());
    ctx->decoded = true;
    ctx->results_valid = false;
    return EFX_OK;
}

int efx_decode(efx_ctx* ctx) { return efx_decode_from(ctx, 0); }

int efx_sync(efx_ctx* ctx)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    return sync_all(ctx);
}

static int fetch_results(efx_ctx* ctx)
{
    if (!ctx->decoded)
        return fail(ctx, EFX_ERR_STATE, "no decode has run");
    if (ctx->results_valid)
        return EFX_OK;
    int r = sync_all(ctx);
    if (r)
        return r;
    const bool ts_input = ctx->last_ts_input;
    ctx->n_streams = ctx->last_n_streams;
    const size_t P = (size_t)ctx->cfg.max_pictures;
    ctx->h_pic_count.resize(ctx->n_streams);
    ctx->h_status.resize(ctx->n_streams);
    ctx->h_call_pos.resize(2 * (size_t)ctx->n_streams);
    if (ts_input)
        ctx->h_pts.resize((size_t)ctx->n_streams * P);
    else
        ctx->h_pts.clear();
    ctx->h_counters = DecodeCounters{};
    // every group of the call left the results of ITS streams in its own hand-over slot
    for (int g = 0; g < ctx->n_groups; g++) {
        const efx_ctx::Group& gr = ctx->groups[g];
        const efx_ctx::Slot& sl = ctx->slot[gr.slot];
        const size_t f = (size_t)gr.first, c = (size_t)gr.count;
        if (!c)
            continue;
        EFX_HIP(hipMemcpy(ctx->h_pic_count.data() + f, sl.d_pic_count + f, c * sizeof(uint32_t), hipMemcpyDeviceToHost));
        EFX_HIP(hipMemcpy(ctx->h_status.data() + f, sl.d_status + f, c * sizeof(uint32_t), hipMemcpyDeviceToHost));
        EFX_HIP(hipMemcpy(ctx->h_call_pos.data() + 2 * f, sl.d_call_pos + 2 * f, 2 * c * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (ts_input)
            EFX_HIP(hipMemcpy(ctx->h_pts.data() + f * P, sl.d_pts + f * P, c * P * sizeof(int64_t), hipMemcpyDeviceToHost));
        for (int h = gr.half0; h < gr.half0 + gr.halves; h++) {
            DecodeCounters dc;
            EFX_HIP(hipMemcpy(&dc, sl.d_counters + h, sizeof(dc), hipMemcpyDeviceToHost));
            ctx->h_counters.total_slices += dc.total_slices;
            ctx->h_counters.coefficients += dc.coefficients;
            ctx->h_counters.macroblocks += dc.macroblocks;
        }
    }
    ctx->results_valid = true;
    return EFX_OK;
}

int efx_picture_count(efx_ctx* ctx, int stream, int* n_pictures)
{
    bind_device(ctx);
    if (!ctx || !n_pictures || stream < 0)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    if (stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    *n_pictures = (int)ctx->h_pic_count[stream];
    return EFX_OK;
}

int efx_stream_status(efx_ctx* ctx, int stream, uint32_t* bits)
{
    bind_device(ctx);
    if (!ctx || !bits || stream < 0)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    if (stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    *bits = ctx->h_status[stream];
    return EFX_OK;
}

int efx_picture_pts(efx_ctx* ctx, int stream, int picture, int64_t* pts)
{
    bind_device(ctx);
    if (!ctx || !pts || stream < 0 || picture < 0)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    if (stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    if (ctx->h_pts.empty()) {
        *pts = picture;  // elementary-stream input carries no PTS: pictures are numbered
        return EFX_OK;
    }
    *pts = picture < (int)ctx->h_pic_count[stream] ? ctx->h_pts[(size_t)stream * ctx->cfg.max_pictures + picture] : -1;
    return EFX_OK;
}

int efx_stream_picture_slot(efx_ctx* ctx, int stream, int picture, int* slot)
{
    bind_device(ctx);
    if (!ctx || !slot || stream < 0 || picture < 0)
        return EFX_ERR_ARG;
    int r = fetch_results(ctx);
    if (r)
        return r;
    if (stream >= ctx->n_streams)
        return EFX_ERR_ARG;
    const int pos0 = ctx->h_call_pos[2 * (size_t)stream], f = ctx->h_call_pos[2 * (size_t)stream + 1];
    const uint32_t q = (uint32_t)pos0 + (uint32_t)(f < 0 ? picture + 1 : std::max(0, picture - f));
    *slot = (int)(q % (uint32_t)ctx->cfg.ring_depth);
    return EFX_OK;
}

int efx_picture_slot(efx_ctx* ctx, int picture)
{
    bind_device(ctx);
    int slot = 0;
    int r = efx_stream_picture_slot(ctx, 0, picture, &slot);
    return r ? r : slot;
}

int efx_frame_device_ptr(efx_ctx* ctx, int stream, int slot, void** dptr)
{
    bind_device(ctx);
    if (!ctx || !dptr || stream < 0 || stream >= ctx->cfg.max_streams || slot < 0 || slot >= ctx->cfg.ring_depth)
        return EFX_ERR_ARG;
    *dptr = ctx->d_frames + ((size_t)stream * ctx->cfg.ring_depth + slot) * kFrameBytes;
    return EFX_OK;
}

int efx_download_frame(efx_ctx* ctx, int stream, int slot, uint8_t* dst)
{
    bind_device(ctx);
    void* p;
    int r = efx_frame_device_ptr(ctx, stream, slot, &p);
    if (r || !dst)
        return r ? r : EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    EFX_HIP(hipMemcpy(dst, p, kFrameBytes, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_upload_frame(efx_ctx* ctx, int stream, int slot, const uint8_t* src)
{
    bind_device(ctx);
    void* p;
    int r = efx_frame_device_ptr(ctx, stream, slot, &p);
    if (r || !src)
        return r ? r : EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    EFX_HIP(hipMemcpy(p, src, kFrameBytes, hipMemcpyHostToDevice));
    return EFX_OK;
}

int efx_frame_hashes(efx_ctx* ctx, int first_stream, int n, uint64_t* out)
{
    bind_device(ctx);
    if (!ctx || !out || first_stream < 0 || n <= 0 || first_stream + n > ctx->cfg.max_streams)
        return EFX_ERR_ARG;
    const int D = ctx->cfg.ring_depth;
    const int frames = n * D;
    hipLaunchKernelGGL(k_frame_hash, dim3((frames + 63) / 64), dim3(64), 0, ctx->stream,
                       ctx->d_frames + (size_t)first_stream * D * kFrameBytes, frames, ctx->d_hash);
    EFX_HIP(hipGetLastError());
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
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
    bind_device(ctx);
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
    const int lines = o->ntsc ? 262 : 312, groups = (o->ntsc ? 912 : 1136) / 8;  // (video_init / pal_init, video.cpp:554-630)
    const int blocks = o->n_streams * ((lines * groups + kCompositeItemsPerBlock - 1) / kCompositeItemsPerBlock);
    hipLaunchKernelGGL(k_composite, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_frames, ctx->d_video[o->ntsc ? 1 : 0],
                       ctx->d_video_lines[o->ntsc ? 1 : 0], a, dst_device);
    EFX_HIP(hipGetLastError());
    return EFX_OK;
}

int efx_composite_fields(efx_ctx* ctx, int first_stream, int n_streams, int slot, int ntsc, int frame_counter,
                         uint16_t* dst_device)
{
    bind_device(ctx);
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
    bind_device(ctx);
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
    bind_device(ctx);
    if (!ctx || !ts || !len || !audio_device || !audio_len_device || n_streams <= 0)
        return fail(ctx, EFX_ERR_ARG, "efx_demux_audio: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_demux_audio: more streams than max_streams");
    int r = ensure_ts_buffers(ctx);
    if (r)
        return r;
    // the TS staging buffers are shared with efx_upload_streams: nothing may be in flight
    r = sync_all(ctx);
    if (r)
        return r;
    uint8_t* const h_stage = ctx->up[0].h_es;
    std::vector<uint64_t> off((size_t)n_streams + 1), out_off((size_t)n_streams + 1);
    std::vector<uint32_t> tlen(n_streams), pbase(n_streams);
    size_t pos = 0, packets = 0, max_len = 0;
    for (int i = 0; i < n_streams; i++) {
        pbase[i] = (uint32_t)packets;
        packets += len[i] / 188;
        max_len = std::max(max_len, len[i]);
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
            memcpy(h_stage + pos, ts[i], len[i]);
        memset(h_stage + pos + len[i], 0, padded - len[i]);
        tlen[i] = (uint32_t)len[i];
        pos += padded;
    }
    off[n_streams] = pos;
    out_off[n_streams] = (uint64_t)n_streams * stride;
    if (packets > ctx->pes_cap)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_demux_audio: more packets than the context's transport-stream capacity");
    hipStream_t st = ctx->stream;
    uint64_t* d_out_off = nullptr;
    EFX_HIP(dalloc(&d_out_off, (size_t)n_streams + 1));
    hipError_t e = hipMemcpyAsync(ctx->d_ts, h_stage, pos, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_ts_off, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out_off, out_off.data(), out_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_idx_len, tlen.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_idx_base, pbase.data(), n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const unsigned chunks = (unsigned)std::max<size_t>(1, (max_len / 188 + kDemuxChunk - 1) / kDemuxChunk);
        hipLaunchKernelGGL(k_demux_audio_scan, dim3(chunks, n_streams), dim3(kDemuxThreads), 0, st, ctx->d_ts, ctx->d_ts_off, ctx->d_idx_len, ctx->d_idx_base,
                           ctx->d_demux_chunks);
        hipLaunchKernelGGL(k_demux_audio_offsets, dim3(n_streams), dim3(64), 0, st, ctx->d_idx_len, ctx->d_idx_base, ctx->d_demux_chunks,
                           audio_len_device);
        hipLaunchKernelGGL(k_demux_audio, dim3(chunks, n_streams), dim3(kDemuxThreads), 0, st, ctx->d_ts, ctx->d_ts_off, ctx->d_idx_len, ctx->d_idx_base,
                           ctx->d_demux_chunks, audio_device, d_out_off);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host staging buffer and d_out_off are free again
    (void)dev_free(d_out_off);
    if (e != hipSuccess)
        return fail(ctx, EFX_ERR_DEVICE, "efx_demux_audio", e);
    return EFX_OK;
}

// ---- trick-play index (indexer/indexer.cpp, espflix.cpp:573-629) -------------------------------------

int efx_index_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* ts, const size_t* len, const uint32_t* trick_speed,
                      uint32_t bin_size, efx_idx_rec* recs, uint32_t* samples, size_t samples_cap)
{
    bind_device(ctx);
    if (!ctx || !ts || !len || !recs || !samples || n_streams <= 0 || bin_size == 0 || samples_cap == 0)
        return fail(ctx, EFX_ERR_ARG, "efx_index_streams: bad argument");
    if (n_streams > ctx->cfg.max_streams)
        return fail(ctx, EFX_ERR_CAPACITY, "efx_index_streams: more streams than max_streams");
    int r = ensure_ts_buffers(ctx);
    if (r)
        return r;
    // the TS staging buffers are shared with efx_upload_streams: nothing may be in flight
    r = sync_all(ctx);
    if (r)
        return r;
    uint8_t* const h_stage = ctx->up[0].h_es;
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
            memcpy(h_stage + pos, ts[i], len[i]);
        memset(h_stage + pos + len[i], 0, padded - len[i]);
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
        (void)dev_free(d_samples);
        return code;
    };
    hipError_t e = hipMemcpyAsync(ctx->d_ts, h_stage, pos, hipMemcpyHostToDevice, st);
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
    bind_device(ctx);
    if (!ctx || !frames_device || !state_device || !pcm_device || n_streams <= 0 || n_frames < 0 || frame_bytes <= 0 ||
        (size_t)frame_bytes * (size_t)n_frames >= (1u << 28) || ((uintptr_t)state_device & 3))
        return EFX_ERR_ARG;
    SbcState* const state = static_cast<SbcState*>(state_device);
    if (n_frames == 0 || ctx->opt_sbc_serial) {
        // one wave per stream walking its frames (no frames: it leaves the state as it reads it and a PCM count of zero)
        hipLaunchKernelGGL(k_sbc, dim3(n_streams), dim3(64), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames, state,
                           ctx->d_sbc_tables, pcm_device, pcm_stride, ret_device, pcm_count_device, flags);
        EFX_HIP(hipGetLastError());
        return EFX_OK;
    }
    // scratch: which kernel decodes a stream + the three work lists + their counters; the states on their way; per frame the
    // bit allocation (k_sbc_frames) and, for streams that are not regular, the plan (k_sbc_plan)
    if ((size_t)n_streams > ctx->sbc_flags_cap) {
        if (ctx->d_sbc_flags)
            (void)dev_free(ctx->d_sbc_flags);
        if (ctx->d_sbc_next)
            (void)dev_free(ctx->d_sbc_next);
        ctx->d_sbc_flags = nullptr;
        ctx->d_sbc_next = nullptr;
        ctx->sbc_flags_cap = 0;
        EFX_HIP(dalloc(&ctx->d_sbc_flags, (size_t)n_streams * 4 + sizeof(SbcQueues) / 4));
        EFX_HIP(dalloc(&ctx->d_sbc_next, (size_t)n_streams));
        ctx->sbc_flags_cap = (size_t)n_streams;
        ctx->sbc_flags_clean = false;
    }
    const size_t info_need = (size_t)n_streams * ((size_t)n_frames + 1);
    if (info_need > ctx->sbc_info_cap) {
        if (ctx->d_sbc_info)
            (void)dev_free(ctx->d_sbc_info);
        if (ctx->d_sbc_plan)
            (void)dev_free(ctx->d_sbc_plan);
        ctx->d_sbc_info = nullptr;
        ctx->d_sbc_plan = nullptr;
        ctx->sbc_info_cap = 0;
        EFX_HIP(dalloc(&ctx->d_sbc_info, info_need));
        EFX_HIP(dalloc(&ctx->d_sbc_plan, info_need));
        ctx->sbc_info_cap = info_need;
    }
    // per (stream, granule of eight frames): a slot in each of the two tables of chunks a regular kernel takes, a cover byte
    const int n_gran = (n_frames + 7) / 8, extra_cap = n_streams * n_gran;
    if ((size_t)extra_cap > ctx->sbc_gran_cap) {
        if (ctx->d_sbc_extra)
            (void)dev_free(ctx->d_sbc_extra);
        if (ctx->d_sbc_cover)
            (void)dev_free(ctx->d_sbc_cover);
        ctx->d_sbc_extra = nullptr;
        ctx->d_sbc_cover = nullptr;
        ctx->sbc_gran_cap = 0;
        EFX_HIP(dalloc(&ctx->d_sbc_extra, 2 * (size_t)extra_cap));
        EFX_HIP(dalloc(&ctx->d_sbc_cover, (size_t)extra_cap));
        ctx->sbc_gran_cap = (size_t)extra_cap;
    }
    uint32_t* const d_how = ctx->d_sbc_flags;
    uint32_t* const d_lists = d_how + ctx->sbc_flags_cap;
    SbcQueues* const d_queues = reinterpret_cast<SbcQueues*>(d_how + 4 * ctx->sbc_flags_cap);
    // parallel[]: every word kSbcRegularFlag when k_sbc_frames starts -- as the previous call's k_sbc_finish left them
    if (!ctx->sbc_flags_clean)
        EFX_HIP(hipMemsetAsync(d_how, 0xFF, ctx->sbc_flags_cap * sizeof(uint32_t), ctx->stream));
    ctx->sbc_flags_clean = false;
    hipLaunchKernelGGL(k_sbc_frames, dim3((n_frames + 255) / 256, n_streams), dim3(256), 0, ctx->stream, frames_device, stream_stride,
                       frame_bytes, n_frames, ctx->d_sbc_info, ret_device, d_how, d_queues);
    hipLaunchKernelGGL(k_sbc_plan, dim3(n_streams + 1), dim3(256), 0, ctx->stream, ctx->d_sbc_info, n_frames, flags, state, ctx->d_sbc_plan,
                       d_how, d_queues, d_lists, n_streams, (int)ctx->sbc_flags_cap, ret_device, pcm_count_device, ctx->d_sbc_extra,
                       ctx->d_sbc_cover);
    // Persistent workgroups take (stream, chunk of frames) items off the list of their kind, dealt round robin; a list that
    // stayed empty costs its kernel one look.  Grids = what is resident at a time: EFX_SBC_WG_PER_CU workgroups per compute unit
    // 
    // for the mono kernel (16 frames per item: 36 KB of LDS), 4 for the other two (35 KB of LDS).
    const long long items = (long long)n_streams * ((n_frames + 8) / 8);
    const int g_mono = (int)std::min<long long>(items, (long long)ctx->n_cus * EFX_SBC_WG_PER_CU), g_wide = (int)std::min<long long>(items, (long long)ctx->n_cus * 4);
    hipLaunchKernelGGL(k_sbc_par_mono, dim3(g_mono), dim3(256), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames, state,
                       ctx->d_sbc_next, ctx->d_sbc_tables, ctx->d_sbc_info, pcm_device, pcm_stride, pcm_count_device, flags, d_queues, d_lists,
                       (int)ctx->sbc_flags_cap, ctx->d_sbc_extra, extra_cap);
    hipLaunchKernelGGL(k_sbc_par_stereo, dim3(g_wide), dim3(256), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames, state,
                       ctx->d_sbc_next, ctx->d_sbc_tables, ctx->d_sbc_info, pcm_device, pcm_stride, pcm_count_device, flags, d_queues, d_lists,
                       (int)ctx->sbc_flags_cap, ctx->d_sbc_extra, extra_cap);
    hipLaunchKernelGGL(k_sbc_gen, dim3(g_wide), dim3(256), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames, state,
                       ctx->d_sbc_next, ctx->d_sbc_tables, ctx->d_sbc_info, ctx->d_sbc_plan, pcm_device, pcm_stride, flags, d_queues, d_lists,
                       (int)ctx->sbc_flags_cap, ctx->d_sbc_cover, n_gran);
    // the new states take their place; a stream whose state no call of this library can have left is decoded by one wave
    hipLaunchKernelGGL(k_sbc_finish, dim3(n_streams), dim3(64), 0, ctx->stream, frames_device, stream_stride, frame_bytes, n_frames, state,
                       ctx->d_sbc_next, ctx->d_sbc_tables, pcm_device, pcm_stride, ret_device, pcm_count_device, flags, d_how);
    EFX_HIP(hipGetLastError());
    ctx->sbc_flags_clean = true;
    return EFX_OK;
}

int efx_set_timing(efx_ctx* ctx, int enable)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    if (enable && ctx->timing_ring.empty()) {
        ctx->timing_ring.resize((size_t)kTimingRing * kMaxGroups);  // (the events of a set are created at its first use)
    }
    ctx->timing = enable != 0;
    ctx->timed_calls = 0;  // (re-)enabling starts a new averaging window
    return EFX_OK;
}

int efx_get_timing(efx_ctx* ctx, efx_timing* out)
{
    bind_device(ctx);
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
        // a call's stage time = the sum over its groups of streams (they run one after the other on their stream)
        const size_t call = (size_t)((ctx->timed_calls - 1 - k) % kTimingRing);
        const int G = ctx->timing_groups[call];
        const uint32_t leaders = ctx->timing_leaders[call];
        int last_leader = 0, n_leaders = 0;
        for (int g = 0; g < G; g++) {
            const efx_ctx::TimingEvents& te = ctx->timing_ring[call * kMaxGroups + g];
            float a = 0, b = 0, c = 0;
            EFX_HIP(hipEventElapsedTime(&a, te.ev[0], te.ev[1]));
            EFX_HIP(hipEventElapsedTime(&b, te.ev[1], te.ev[2]));
            if (leaders >> g & 1) {
                EFX_HIP(hipEventElapsedTime(&c, te.ev[4], te.ev[3]));
                last_leader = g;
                n_leaders++;
            }
            t.index_ms += a / n_timed;
            t.parse_ms += b / n_timed;
            t.recon_ms += c / n_timed;
        }
        float d = 0;
        EFX_HIP(hipEventElapsedTime(&d, ctx->timing_ring[call * kMaxGroups].ev[0], ctx->timing_ring[call * kMaxGroups + last_leader].ev[3]));
        t.total_ms += d / n_timed;
        if (k == 0) {  // the launch structure of the newest call
            t.groups = (uint32_t)n_leaders;
            t.parse_halves = (uint32_t)G;
            t.recon_launches = ctx->timing_launches[call];
        } else if (t.groups != (uint32_t)n_leaders || t.parse_halves != (uint32_t)G || t.recon_launches != ctx->timing_launches[call])
            t.mixed = 1;
    }
    t.timed_calls = (uint32_t)n_timed;
    for (int i = 0; i < ctx->n_streams; i++)
        t.pictures += ctx->h_pic_count[i];
    t.slices = ctx->h_counters.total_slices;
    t.coefficients = ctx->h_counters.coefficients;
    {
        efx_ctx::Upload& u = ctx->up[ctx->last_upload];
        t.es_bytes = u.es_used;
        t.ts_bytes = u.ts_bytes;
        if (u.ts_input && u.demux_timed)
            EFX_HIP(hipEventElapsedTime(&t.demux_ms, u.ev_demux[0], u.ev_demux[1]));
    }
    *out = t;
    return EFX_OK;
}


// EFX_GUARD's own check (tools/guard_selftest.py): one word written at `offset_bytes` from the start of a device buffer
// (negative: in front of it).  Under the guard-page allocator the first 16-byte unit behind the buffer (mode 1) / any byte in
// front of it (mode 2) faults; without it the write lands in whatever lies there -- never call this outside that test.
int efx_debug_poke(efx_ctx* ctx, void* dptr, size_t bytes, long long offset_bytes)
{
    bind_device(ctx);
    if (!ctx || !dptr)
        return EFX_ERR_ARG;
    (void)bytes;
    char* at = static_cast<char*>(dptr) + offset_bytes;
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<uint32_t*>(at), 0xDEADBEEFu, (size_t)1);
    EFX_HIP(hipGetLastError());
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    return EFX_OK;
}

// development: the header lines of k_recon_all's hand-over words after the most recent call (word 0 of each of the 64 lines;
// lines 16 ... carry per-phase times only in an -DEFX_RA_STATS build: tools/exp/r5_recon_check.py)
int efx_debug_recon_stats(efx_ctx* ctx, uint32_t out[64])
{
    bind_device(ctx);
    if (!ctx || !out || !ctx->decoded || ctx->n_groups < 1)
        return EFX_ERR_ARG;
    int r = sync_all(ctx);
    if (r)
        return r;
    std::vector<uint32_t> all(64 * 32);
    EFX_HIP(hipMemcpy(all.data(), ctx->slot[ctx->groups[ctx->n_groups - 1].slot].d_recon_sync, all.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < 64; k++)
        out[k] = all[(size_t)k * 32];
    return EFX_OK;
}

int efx_device_alloc(efx_ctx* ctx, size_t bytes, void** dptr)
{
    bind_device(ctx);
    if (!ctx || !dptr)
        return EFX_ERR_ARG;
    EFX_HIP(dev_alloc(dptr, bytes));
    return EFX_OK;
}

int efx_device_free(efx_ctx* ctx, void* dptr)
{
    bind_device(ctx);
    if (!ctx)
        return EFX_ERR_ARG;
    EFX_HIP(dev_free(dptr));
    return EFX_OK;
}

int efx_memcpy_h2d(efx_ctx* ctx, void* dst_device, const void* src, size_t bytes)
{
    bind_device(ctx);
    if (!ctx || !dst_device || !src)
        return EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    EFX_HIP(hipMemcpy(dst_device, src, bytes, hipMemcpyHostToDevice));
    return EFX_OK;
}

int efx_memcpy_d2h(efx_ctx* ctx, void* dst, const void* src_device, size_t bytes)
{
    bind_device(ctx);
    if (!ctx || !dst || !src_device)
        return EFX_ERR_ARG;
    EFX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->recon_groups_unsynced = 0;  // (the host knows the reconstruction stream idle)
    EFX_HIP(hipMemcpy(dst, src_device, bytes, hipMemcpyDeviceToHost));
    return EFX_OK;
}

}  // extern "C"
