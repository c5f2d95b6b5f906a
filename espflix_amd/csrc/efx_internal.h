// efx_internal.h -- structures shared by the host side of libefx and its gfx950 kernels.
//
// Pipeline of one efx_decode():
//   k_demux        (TS input, at upload) one workgroup per stream: 188-byte packets -> ES + PES PTS list
//                                                                            (player.cpp:294-307,381-493)
//   k_index        one workgroup per stream: start-code scan + header parse + the sequential state rules as scans
//                                                                            (player.cpp:1355-1367,646-730)
//   k_slice_scan   prefix sum of slice counts in picture-major order
//   k_slice_emit   dense slice descriptors
//   k_parse        one lane per slice, a token machine (parse_tm.h): VLC parse -> macroblock records + stream words
//                                                                            (player.cpp:1238-1316,999-1107,891-920)
//   k_recon        one lane per 8x8 block, one launch per picture index: dequantisation + IDCT + half-pel
//                  motion compensation + clamp + strip-layout store          (player.cpp:1110-1121,922-996,732-889,1151-1236)
#pragma once
#include <cstdint>

namespace efx {

constexpr int kMbW = 22, kMbH = 12, kMbCount = 264;
constexpr int kStride = 528, kStripBytes = 8448, kFrameBytes = 101376;
constexpr int kMaxSlicesPerPicture = 16;   // slice start codes kept per picture
#ifndef EFX_PARSE_LANES
#define EFX_PARSE_LANES 64
#endif
constexpr int kParseLanes = EFX_PARSE_LANES;  // slices per k_parse wave (one lane each)
#ifndef EFX_PARSE_WAVES
#define EFX_PARSE_WAVES 4
#endif
constexpr int kParseWaves = EFX_PARSE_WAVES;  // waves of a k_parse workgroup (they share one copy of the tables in LDS)
constexpr int kIndexWaves = 4;             // waves of a k_index workgroup (one workgroup per stream)
constexpr int kMaxUnitsPerStream = 4096;   // start codes indexed per stream per decode
constexpr int kCoefsPerEsByte = 4;         // stream words a slice can need per byte: < 3 (a run/level costs >= 3 bits, a motion
                                           // code's word >= 3 with its share of the macroblock header); 4 keeps every slice's region
                                           // 16-byte aligned for k_parse's staged 16-byte stores
constexpr int kEsTailBytes = 9;            // 00 | 00 00 01 B7 | 00 00 01 B7   (player.cpp:456,472)
constexpr int kEsGuardBytes = 512;         // zero guard after the last stream (parse lanes read 128 B ahead)

// per (stream, picture)
struct PicInfo {
    uint32_t first_slice;  // index into the stream's temporary slice list
    uint16_t n_slices;
    uint8_t type;          // 1 = I, 2 = decoded with the P books (P, and B/D headers the reference ignores)
    uint8_t full_pel;
    uint8_t r_size;        // forward_f_code - 1
    uint8_t custom_q;      // 1: quantiser tables at qtab[(stream*max_pictures+pic)*64]
    uint16_t reserved;
    uint32_t seq_off;      // stream-relative offset of the governing sequence header payload
    uint32_t start_off;    // stream-relative offset of the first byte after the picture start code
};

// one PES header that carried a PTS (k_demux): ES offset of its first payload byte, 33-bit PTS
struct PesEntry {
    uint32_t es_off;
    uint32_t reserved;
    int64_t pts;
};

// decoder state that survives from one efx_decode call to the next (MpegDecoder::_fb_index, _last_pts != -1,
// _pts; reference src/player.h:37-47); efx_reset puts back the constructor's values
struct StreamState {
    uint32_t fb_index;  // frame index of the next picture's predecessor: pictures so far that swapped buffers, + 1
    uint32_t pts_seen;  // a picture header has latched a PES PTS
    int64_t pts_carry;  // newest PES PTS of the uploads so far (-1: none)
};

constexpr int64_t kNoPts = INT64_MIN;  // "this upload carried no PES PTS for the stream" (k_index -> k_advance)

struct SliceTmp {
    uint32_t off;       // stream-relative offset of the first byte after the slice start code
    uint32_t len_code;  // (bytes up to the next start code) << 8 | slice start code value
};

struct SliceDesc {
    uint32_t es_off;   // absolute offset in the ES buffer
    uint32_t es_len;   // bytes up to the next start code
    uint32_t stream;
    uint32_t pic_code_flags;  // pic | code << 8 | type << 16 | full_pel << 18 | r_size << 19 | custom_q << 22
    uint32_t mb_limit;  // the slice stops here: the nearest first macroblock, in raster order, of any other slice of the
                        // picture (264 if none; 0 for a slice superseded by a later one with the same start code) -- a
                        // (damaged) slice that runs on never writes a macroblock another parse lane writes (k_slice_emit)
    uint32_t reserved;
};

// one macroblock of one picture, written by k_parse, consumed by k_recon
struct MbRec {
    uint32_t coef_base;  // absolute index of the first coefficient entry
    uint8_t cnt[6];      // entries per block (intra: including the DC entry)
    uint8_t flags;       // bit0 intra, bit1 skipped (copy co-located), bits 2-6 quantiser_scale, bit7 loaded matrices
    uint8_t epoch;       // decode epoch that wrote the record (0 = never)
    int16_t mvx, mvy;    // half-pel luma displacement after full_pel scaling
};
static_assert(sizeof(MbRec) == 16, "MbRec must be 16 bytes");

// coefficient entry = the parser's stream word: raw code bits << 16 | table value << 6 | scan position, level by
// tm_level() (parse_tm.h); an intra block's first entry: DC value << 6.  k_recon dequantises and pre-multiplies (the
// reference's b[zz] = v * scale_dct_q[zz], player.cpp:1110-1121)

// scan / quantiser table of k_index and k_recon (the VLC tables of the slice parser are parse_tm.h's TmTables)
struct ParseTables {
    uint32_t scan[64];       // scan position n -> zz | slot in k_recon's paired block layout << 8 | default intra q << 16 | 16 << 24
};

struct DecodeCounters {
    uint32_t total_slices;
    uint32_t streams;  // streams of the group of streams these counters belong to (k_slice_scan)
    unsigned long long coefficients;
    unsigned long long macroblocks;
    uint32_t next_wave;  // k_parse: the next group of kParseLanes slices nobody has taken yet (its waves pull their work)
    uint32_t reserved;
};

// composite video geometry + tables (video.cpp:514-630)
struct VideoTables {
    int32_t line_width, line_count, hsync, hsync_long, hsync_short, burst_start, burst_width, active_start;
    int32_t pal;
    uint16_t sync_level, blanking_level, black_level, pad;
    int16_t burst0[64], burst1[64];
    uint32_t color_tab[768];
};

// Every sample of a field that is not picture or overlay: whole lines as video_isr leaves them in
// the DMA buffers (sync, burst, black; vertical blanking; the PAL half-line syncs), one template per
// line kind.  k_composite copies 16 bytes from here instead of re-deriving eight samples.
constexpr int kCompositeItemsPerBlock = 2048;  // 8-sample groups one k_composite workgroup renders (8 passes of 256 threads)
constexpr int kLineTemplates = 6;      // 0/1 normal line (PAL: _line_counter odd / even), 2 NTSC vertical blanking, 3..5 PAL sync types 0, 2, 3
constexpr int kLineTemplateWidth = 1152;  // >= 1136 samples, 16-byte multiple
struct VideoLineTemplates {
    uint16_t tpl[kLineTemplates][kLineTemplateWidth];
};
void build_video_line_templates(const VideoTables* v, VideoLineTemplates* out);

// per-stream result of k_ts_sequences
struct IdxInfo {
    int64_t first_pts, last_pts;  // origin (PTS of the first sequence start), PTS of the last video PES
    uint32_t n_seq;
    uint32_t fast;                // list sorted and short enough for the binary search
};

// k_composite launch arguments (by value)
struct FieldArgs {
    int first_stream, ring_depth;
    int slot, other_slot;   // ring slots of the displayed frame and of the frame scrolled in
    int frame_counter;      // dither phase
    int hscroll;            // multiple of 8 in (-352, 352)
    const uint8_t* overlay; // device, 16 x 80 bytes per stream (or shared: stride 0); may be null
    size_t overlay_stride;
    int overlay_scale;      // 0 = overlay off; 63 full, (63 * blend) >> 5 while fading
    int overlay_progress;
};

// Packets of a transport stream a k_demux workgroup takes, and its threads (k_demux.hip); a multiple of four packets: 4 x 188
// bytes = 47 x 16.  16 packets x one wave: the benchmark's 253-packet streams as 16 k small workgroups instead of 2 k of 128
// packets x 256 threads, whose few long phases left the chip waiting (gather 81 -> 42 us; profiles/r5_demux.md).
#ifndef EFX_DEMUX_CHUNK
#define EFX_DEMUX_CHUNK 16
#endif
constexpr int kDemuxChunk = EFX_DEMUX_CHUNK;
#ifndef EFX_DEMUX_THREADS
#define EFX_DEMUX_THREADS 64
#endif
constexpr int kDemuxThreads = EFX_DEMUX_THREADS;
constexpr uint32_t kDemuxFailedFlag = 0x80000000u;  // in a stream's PES count: k_demux_fused gave up on it (k_index: EFX_STREAM_INTERNAL)
static_assert(kDemuxChunk % 4 == 0 && kDemuxChunk >= 4 && kDemuxChunk <= 256, "k_demux stages whole 16-byte groups");

// SBC synthesis tables (sbc_decoder.cpp:41-71), generated from the A2DP definitions
struct SbcTables {
    int32_t syn[128];   // syn[i * 8 + k] = floor(65536 cos((i + 4)(2k + 1) pi / 16))
    int32_t proto[80];  // proto[i * 10 + j] = floor(-8 * 32768 * Proto_8_80[8 j + i])
    // IQUANT's C division by 2^bits - 1 (sbc_decoder.cpp:263-270) as a multiplication: the round-up constant
    // m' = floor(2^32 (2^l - d) / d) + 1 of Granlund & Montgomery's unsigned division (l = bits; d = 1: m' = 1, no shifts):
    // a / d = (t + ((a - t) >> 1)) >> (bits - 1), t = mulhi(m', a), for every 32-bit a
    uint32_t iq_magic[20];
};

// What k_sbc_frames leaves per (stream, frame): the header's verdict and the bit allocation -- one thread per frame with all
// 64 lanes of a wave busy, instead of once per workgroup that has the frame in reach.
struct alignas(16) SbcFrameInfo {
    uint8_t bits[2][8];       // bits of a sample of (channel, subband)
    uint8_t prefix[2][8];     // ... and where it starts inside a block's bits
    uint16_t per_block;       // bits of one block, all channels
    uint16_t framelen;        // get_samples()'s return value (0xFFFF = -1)
    uint8_t h1, bitpool, flags, pad;
    uint32_t pad2[2];
};
static_assert(sizeof(SbcFrameInfo) == 48, "SbcFrameInfo is moved as three uint4");
constexpr uint32_t kSbcSync = 1, kSbcOk = 2;  // flags: sync byte and length in order (the header moves the geometry); decodable

// What k_sbc_plan leaves per (stream, VIRTUAL frame) of a stream that is not regular (virtual frame v = frame max(v - probe, 0):
// decode_audio()'s frame-size probe decodes frame 0 once more up front).  Everything the reference chains from frame to frame
// -- the geometry a rejected frame is synthesised under, the stale samples it synthesises, where its PCM goes, where its rows
// sit on a channel's timeline of matrixing outputs -- as prefix scans, so that any chunk of frames can be decoded on its own.
struct alignas(16) SbcFramePlan {
    int32_t src[8];     // [q * 2 + c]: the frame whose get_samples() last wrote blocks 4q .. 4q+3 of channel c; -1 = the state's
    uint32_t pcm_off;   // PCM samples of this call before this frame's
    uint32_t vb[2];     // rows on channel c's timeline before this frame's
    int32_t back[2];    // the latest earlier frame with rows on channel c's timeline; -1 = none in this call
    int32_t gsrc;       // the frame whose header left the geometry this frame is synthesised under; -1 = the state's
    uint32_t geom;      // blocks | channels << 8 | synthesised << 16
    uint32_t pad;       // (k_sbc_plan's own: the first frame of the run of frames decoding under one geometry that this frame ends)
};
static_assert(sizeof(SbcFramePlan) == 64, "SbcFramePlan is moved as four uint4");

// efx_sbc_decode's work lists: streams by the kernel that decodes them (kSbcMono / kSbcStereo: regular streams -- every frame
// decodes, one geometry --, kSbcGeneral: the rest), filled by k_sbc_plan, drained by persistent workgroups
constexpr int kSbcMono = 0, kSbcStereo = 1, kSbcGeneral = 2;
struct SbcQueues {
    uint32_t count[4];  // streams on list c
    uint32_t extra[4];  // kSbcMono, kSbcStereo: single chunks of GENERAL streams that a regular kernel can take (SbcExtraItem)
};
// A chunk of a general stream whose frames -- and the frames that hold the nine blocks before it -- all decode under one
// geometry is, but for where its PCM goes, a regular stream's: k_sbc_plan hands it to the regular kernel of its kind.
struct alignas(16) SbcExtraItem {
    uint32_t entry;     // stream | block-count code << 30 (as the list entries of regular streams)
    uint32_t chunk;     // in the kernel's own chunk size
    uint32_t pcm_base;  // PCM samples of the call before the chunk's first frame
    uint32_t pad;
};

// per-stream SBC decoder state (the reference's SBC_Decode, sbc_decoder.h:12-25, with the sliding
// synthesis buffer kept as the last nine rows of matrixing outputs); all zero = sbc_init()
struct SbcState {
    int32_t hist[2][9][16];
    int32_t sb_sample[16][2][8];
    uint8_t frequency, blocks, channels, mode, allocation, subbands, bitpool, reserved;
    uint8_t pad[8];
};
static_assert(sizeof(SbcState) == 2192, "SbcState layout");

void build_sbc_tables(SbcTables* t);
void build_parse_tables(ParseTables* t);
void build_video_tables(int ntsc, VideoTables* t);

}  // namespace efx
