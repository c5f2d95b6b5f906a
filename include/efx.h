/* efx.h -- C-ABI of libefx: batched MPEG-1 video decode, composite-video line synthesis and
 * PDM modulation on AMD Instinct MI355X (gfx950).
 *
 * This is the drop-in boundary for the espflix hot path.  The reference (rossumur/espflix) has
 * no FFI layer: its host player talks to a C++ class and a handful of free functions.  Each
 * entry point below names the reference interface it stands in for (file:line under the
 * reference tree); include/efx_player.hpp re-declares that C++ surface (Frame, MpegDecoder,
 * push_video, video_isr, write_pcm_16) on top of this header for batch = 1.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (EFX_OK) or a negative
 * efx_status; nothing throws; one context per host thread (thread-compatible).  Process-global state: none that a
 * caller can observe -- one mutex-guarded pool of parked HIP streams per device, so that a context created after
 * another was destroyed runs on the same hardware queues (streams are never destroyed).
 * Every entry point makes its context's device current for the calling thread (hipSetDevice) and leaves it so.
 * All work is queued on the context's HIP streams; efx_sync() waits for it.
 */
#ifndef EFX_H
#define EFX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFX_FRAME_WIDTH 352          /* FB_WIDTH,  src/video.h:30 */
#define EFX_FRAME_HEIGHT 192         /* FB_HEIGHT, src/video.h:31 */
#define EFX_FRAME_STRIDE 528         /* FB_STRIDE, src/video.h:32: 352 luma + 176 chroma bytes */
#define EFX_STRIP_ROWS 16            /* FB_SLICE_HEIGHT, src/video.h:33 */
#define EFX_STRIPS 12                /* FB_SLICES, src/video.h:34 */
#define EFX_STRIP_BYTES 8448         /* 16 x 528 */
#define EFX_FRAME_BYTES 101376       /* 12 x 8448; class Frame, src/video.h:36-44 */

typedef enum efx_status {
    EFX_OK = 0,
    EFX_ERR_ARG = -1,           /* null / out-of-range argument */
    EFX_ERR_DEVICE = -2,        /* HIP runtime error (efx_last_error has the text) */
    EFX_ERR_NO_DEVICE = -3,     /* no gfx950 device / device index out of range */
    EFX_ERR_CAPACITY = -4,      /* more streams / bytes / pictures than the context was sized for */
    EFX_ERR_STATE = -5,         /* call made in the wrong order (e.g. decode before upload) */
    EFX_ERR_STREAM = -6         /* a stream violated a constraint; see efx_stream_status */
} efx_status;

/* per-stream status bits reported by efx_stream_status() */
#define EFX_STREAM_OK 0u
#define EFX_STREAM_BAD_SIZE 1u        /* sequence header is not 352x192 (frame store is fixed) */
#define EFX_STREAM_TRUNCATED 2u       /* more pictures than max_pictures: the rest was ignored */
#define EFX_STREAM_TOO_MANY_UNITS 4u  /* start-code index overflow */
#define EFX_STREAM_BAD_VLC 8u         /* an invalid code ended a slice early, or a slice's codes ran through the next start code */
#define EFX_STREAM_MB_OVERRUN 16u     /* a slice ran past the last macroblock row, or into the macroblocks of the next slice */
#define EFX_STREAM_COEF_OVERRUN 32u   /* a block ran past 64 coefficients (block dropped) */
#define EFX_STREAM_SERIAL_HUNT 64u    /* bits the reference's marker hunt (player.cpp:1360-1363: skip zero bits, DISCARD 24 bits, \
                                         take 8 as the marker) would misread: non-zero bits after a header it ignores (picture \
                                         types other than I / P, player.cpp:710-717), a user_data / extension payload that is not \
                                         made of harmless 4-byte groups (player.cpp:1328-1330), bytes ahead of the first start \
                                         code; also a slice start code ahead of the first picture header (the reference parses \
                                         it with the P books and the constructor's state; dropped here).  The reference then \
                                         acts on phantom markers; this decoder indexes byte-aligned start codes, so its output \
                                         for the stream is NOT the reference's */
#define EFX_STREAM_SLICE_ORDER 128u   /* a picture's slice start codes do not rise strictly in bitstream order (a row coded \
                                         twice, rows out of raster order).  Every slice is parsed by its own lane and stops \
                                         where any other slice of the picture starts; of slices with the same code only the \
                                         last is parsed.  The reference, one serial decoder, lets whatever comes LATER in the \
                                         bitstream overwrite: the same frames when every such slice is complete, not when one \
                                         of them is also damaged or short -- the parity claim does not cover a stream with \
                                         this bit */

#define EFX_STREAM_INTERNAL 256u      /* never expected: a reconstruction wave gave up waiting for the stream's previous picture \
                                         (k_recon_all's hand-over counter, optional EFX_OPT_RECON_MODE only; patience: 10 s \
                                         of wall-clock time) -- the launch was abandoned: NO frame of that call is valid, \
                                         for this stream or any other */

typedef enum efx_format {
    EFX_FORMAT_ES = 0, /* raw ISO 11172-2 video elementary stream */
    EFX_FORMAT_TS = 1  /* 188-byte transport packets, video on PID 0x100 (src/player.cpp:381-493);
                          demultiplexed on the device at upload (k_demux), PES PTS kept per picture */
} efx_format;

typedef struct efx_config {
    int device;              /* HIP device ordinal */
    int max_streams;         /* batch capacity */
    int max_pictures;        /* pictures per stream per efx_decode() call */
    int ring_depth;          /* frames kept per stream: 2 = the reference's double buffer
                                (MpegDecoder::_fb, src/player.h:37-40); max_pictures+1 keeps all.  A macroblock no slice of
                                its picture covers (a slice missing from the stream) keeps what its ring slot held: with 2
                                that is the picture two back, as in the reference; with a deeper ring something older --
                                pictures with missing slices match the reference only at ring_depth 2 */
    size_t max_stream_bytes; /* total ES bytes per upload (0 = 16 KiB x pictures x streams) */
    void* hip_stream;        /* hipStream_t to run on, or NULL for a private stream */
} efx_config;

typedef struct efx_ctx efx_ctx;

/* -- lifetime --------------------------------------------------------------------------- */
/* MpegDecoder::MpegDecoder + Frame::init (src/player.cpp:354-369,25-31): allocates the frame
 * rings (zero filled) and all scratch in HBM. */
int efx_create(const efx_config* cfg, efx_ctx** out);
void efx_destroy(efx_ctx* ctx);
const char* efx_last_error(const efx_ctx* ctx);
const char* efx_status_string(int status);

/* -- bitstream in ------------------------------------------------------------------------ */
/* Stands in for the Buffer hand-off MpegDecoder::push_full / pop_empty (src/player.cpp:371-379,
 * src/streamer.h:139-143) for a whole batch: copies n_streams byte ranges (host pointers, the
 * caller keeps ownership) into HBM.  EFX_FORMAT_TS input (188-byte packets, video on PID 0x100)
 * is demultiplexed ON THE DEVICE: MpegDecoder::more/demux/parse_pts (src/player.cpp:294-307,
 * 381-436,459-493) for the whole batch -- adaptation fields and PES headers skipped at the
 * reference's fixed offsets, one zero byte per packet that lost sync, PES PTS values kept for
 * efx_picture_pts.  Like MpegDecoder::more() at end of data (src/player.cpp:456,469-473) each
 * stream is terminated with 00 | 00 00 01 B7 | 00 00 01 B7.  Does NOT reset the frame rings.
 * Three bitstream buffers take turns: the call copies into pinned staging memory and queues the
 * transfer (and k_demux) on a copy stream, then returns; the GPU may still be decoding the previous
 * batches, so ingest and decode of consecutive batches overlap.  efx_decode decodes the batch uploaded
 * last. */
int efx_upload_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* data, const size_t* len, int format);

/* In-place ingest: the reference's contract is "the caller owns the Buffer, the decoder reads it where it lies"
 * (class Buffer, src/streamer.h:139-143; filled by the app in decode_next, src/espflix.cpp:723-737, read by
 * MpegDecoder::more, src/player.cpp:459-493).  Its batch form: an ARENA of page-locked, device-visible host memory
 * (efx_host_alloc, or the caller's own memory made so by efx_host_register) in which the caller lays the streams of
 * a batch out the way the device buffer holds them -- stream i at offsets[i] of efx_stream_layout(), i.e. 16-byte
 * aligned starts with room behind every stream for the end-of-data tail.  efx_upload_streams_inplace takes such a batch
 * (every data[i] == data[0] + offsets[i], all of it inside one arena; anything else: EFX_ERR_ARG): it writes the tails and
 * the zero fill INTO the gaps the layout leaves (the only bytes of the arena the library ever writes -- hence the
 * non-const pointers), and the H2D transfer reads the caller's memory directly -- no staging copy, one transfer.  The
 * caller keeps ownership throughout and must leave the batch's bytes alone until efx_upload_done() says the transfer no
 * longer reads them.  In-place ingest is asked for BY NAME: efx_upload_streams() never writes through its `const` data
 * pointers and always takes the staged path, wherever the streams lie (round 5 inferred it from the pointer pattern). */
int efx_upload_streams_inplace(efx_ctx* ctx, int n_streams, uint8_t* const* data, const size_t* len, int format);
int efx_host_alloc(efx_ctx* ctx, size_t bytes, void** host_ptr);
int efx_host_free(efx_ctx* ctx, void* host_ptr);
int efx_host_register(efx_ctx* ctx, void* host_ptr, size_t bytes);
int efx_host_unregister(efx_ctx* ctx, void* host_ptr);
/* offsets[0 .. n_streams]: where stream i of the given lengths starts inside an arena (offsets[0] = 0), offsets[n_streams] =
 * bytes the batch occupies.  Host only. */
int efx_stream_layout(int n_streams, const size_t* len, size_t* offsets);
/* 1: the most recent efx_upload_streams[_inplace] no longer reads caller memory (always so for the staged path once the
 * call has returned); 0: its transfer is still in flight; negative: error. */
int efx_upload_done(efx_ctx* ctx);
/* The elementary stream the decoder sees for `stream` (what MpegDecoder::more() feeds the bit
 * reader, src/player.cpp:459-493), without the end-of-data tail: *es_len receives its length,
 * up to `cap` bytes are copied to dst (dst may be NULL when cap is 0). */
int efx_download_es(efx_ctx* ctx, int stream, uint8_t* dst, size_t cap, size_t* es_len);

/* A freshly constructed decoder: Frame::init (src/player.cpp:25-31) zeroes the frame rings, and the
 * per-stream state that otherwise survives from one efx_decode to the next -- the frame index
 * (_fb_index), "a picture has latched a PTS" (_last_pts != -1) and the newest PES PTS (_pts) -- goes
 * back to the constructor's values (src/player.cpp:354-361).  (MpegDecoder::reset(), 439-453, keeps
 * _fb_index and _pts across plays; call efx_reset only where a new MpegDecoder would be made.) */
int efx_reset(efx_ctx* ctx);
/* MpegDecoder::reset() between two plays on the same decoder (src/player.cpp:439-453): _last_pts = -1,
 * i.e. the next picture headers neither push nor swap until one latches a PES PTS again; the frame index,
 * the newest PES PTS and the frame contents survive.  Synchronous. */
int efx_play_reset(efx_ctx* ctx);
/* The per-stream decoder state (valid for what has been queued so far; synchronises): the frame index
 * (MpegDecoder::_fb_index: the next picture is reconstructed into slot frame_index % ring_depth if no buffer
 * swap precedes it, frame_index + 1 otherwise, from the slot before), whether a picture has latched a PTS
 * since the last (play) reset, and the newest PES PTS seen (-1: none).  Any pointer may be NULL. */
int efx_stream_state(efx_ctx* ctx, int stream, uint32_t* frame_index, int* pts_seen, int64_t* newest_pts);
/* Frame::erase (src/player.cpp:48-52): fill every ring frame with 0x30. */
int efx_erase_frames(efx_ctx* ctx);

/* -- decode ------------------------------------------------------------------------------ */
/* MpegDecoder::run() over the uploaded batch (src/player.cpp:1355-1367 and everything below
 * it: marker/sequence/gop/picture/slice/block/idct/mocomp).  Asynchronous.  The decoder keeps
 * going from call to call like the reference fed Buffer after Buffer: every stream carries its
 * frame index and its "a PTS has been latched" state (flush_picture, src/player.cpp:692-702).
 * With p the frame index before the call (1 after efx_reset) picture i of the call is
 * reconstructed into ring slot (p + s(i)) % ring_depth from slot (p + s(i) - 1) % ring_depth,
 * s(i) = i + 1 once a picture has latched a PES PTS, else max(0, i - f) with f the first picture of
 * the call that does: the reference does not swap its two buffers before the first PTS.  Elementary-
 * stream input counts every picture as carrying one.  ring_depth 2 is the reference's pair.
 * Streams are independent; a large batch decoded while the GPU is otherwise idle runs as several groups of
 * streams one after the other (the call pipelines inside itself, efx_timing::groups), back-to-back calls as one
 * group each -- the results do not depend on it. */
int efx_decode(efx_ctx* ctx);
/* The same, starting at picture `first_picture` of every uploaded stream (earlier pictures are walked
 * for their header state only): a stream with more than max_pictures pictures (EFX_STREAM_TRUNCATED)
 * is decoded by efx_decode_from(ctx, 0), (ctx, max_pictures), (ctx, 2 * max_pictures) ... */
int efx_decode_from(efx_ctx* ctx, int first_picture);
/* The same for at most n_pictures (1 ... max_pictures) pictures per stream: a caller that knows its batch holds few
 * pictures (the streaming adapter decoding a real-time play picture by picture, efx_player.hpp) pays for that many
 * reconstruction launches, not for max_pictures of them; streams with more are flagged EFX_STREAM_TRUNCATED as above. */
int efx_decode_range(efx_ctx* ctx, int first_picture, int n_pictures);
int efx_sync(efx_ctx* ctx);

/* number of pictures found in a stream by the last efx_decode (valid after efx_sync) */
int efx_picture_count(efx_ctx* ctx, int stream, int* n_pictures);
/* OR of EFX_STREAM_* bits for a stream (valid after efx_sync) */
int efx_stream_status(efx_ctx* ctx, int stream, uint32_t* bits);
/* PES PTS latched for picture `picture` (flush_picture, src/player.cpp:692-702; -1 when no PES
 * with a PTS preceded it); valid after efx_decode for TS input; ES input yields the picture index. */
int efx_picture_pts(efx_ctx* ctx, int stream, int picture, int64_t* pts);

/* -- frames out (the push_video up-call surface, src/video.h:49) ------------------------- */
/* Ring slot that holds picture `picture` of stream `stream` after the last efx_decode (valid after
 * efx_sync; see efx_decode for the arithmetic). */
int efx_stream_picture_slot(efx_ctx* ctx, int stream, int picture, int* slot);
/* The same for stream 0 as a return value (negative = efx_status): every stream of a batch shares it as
 * long as all of them decoded the same number of pictures in every call since efx_reset. */
int efx_picture_slot(efx_ctx* ctx, int picture);
/* Device pointer to a ring frame in the reference strip layout (12 x 8448 bytes). */
int efx_frame_device_ptr(efx_ctx* ctx, int stream, int slot, void** dptr);
/* Copy one ring frame to host memory (EFX_FRAME_BYTES). Synchronous. */
int efx_download_frame(efx_ctx* ctx, int stream, int slot, uint8_t* dst);
/* FNV-1a-64 of ring frames [first_stream, first_stream+n) x [0, ring_depth), computed on the
 * device, written to host memory as n x ring_depth uint64.  Synchronous. */
int efx_frame_hashes(efx_ctx* ctx, int first_stream, int n, uint64_t* out);
/* Overwrite one ring frame from host memory (tests, poster upload). Synchronous. */
int efx_upload_frame(efx_ctx* ctx, int stream, int slot, const uint8_t* src);

/* -- composite video out (video_init / video_isr, src/video.cpp:572-630,1122-1198) -------- */
typedef struct efx_video_params {
    int line_width, line_count;        /* samples per line, lines per field */
    int hsync, hsync_long, hsync_short;
    int burst_start, burst_width, active_start;
} efx_video_params;
/* geometry video_init(ntsc) establishes: NTSC 912 x 262, PAL 1136 x 312 */
int efx_video_get_params(int ntsc, efx_video_params* out);
/* One field per selected stream: n_streams x line_count x line_width uint16 DAC words into
 * dst (device memory), from ring slot `slot` of streams first_stream..+n.  frame_counter
 * supplies the dither phase (_frame_counter & 1, src/video.cpp:701).  Asynchronous. */
int efx_composite_fields(efx_ctx* ctx, int first_stream, int n_streams, int slot, int ntsc, int frame_counter,
                         uint16_t* dst_device);

/* The same with the two display features of video_isr that involve more than the front frame:
 *  - hscroll: the ease-in / ease-out slide between the two Frames of a pair (_hscroll,
 *    src/video.cpp:1077-1088,1146-1154): a multiple of 8 in (-352, 352); h > 0 shows `slot` from
 *    column h followed by `other_slot` from column 0, h < 0 shows `other_slot` from column
 *    352 + h followed by `slot` from column 0;
 *  - the 80 x 16 time / progress-bar overlay (composite(), src/video.cpp:838-887; the buffer
 *    _video_composite, src/video.h:52-55) on the sixteen lines starting two lines below the
 *    picture: overlay_blend is _video_composite_blend (0 off, -1 or >= 32 full, 1..31 fading --
 *    the caller decrements it once per field as video_isr does, src/video.cpp:1192-1193),
 *    overlay_progress is _video_composite_progress (0..240).
 * overlay (device memory) holds n_streams blocks of 1280 bytes overlay_stride apart, or one
 * block shared by all streams when overlay_stride is 0; NULL with a non-zero blend draws the
 * bar over an all-zero text area. */
typedef struct efx_field_opts {
    int first_stream, n_streams;
    int slot, other_slot;
    int ntsc;
    int frame_counter;
    int hscroll;
    const uint8_t* overlay;
    size_t overlay_stride;
    int overlay_blend;
    int overlay_progress;
} efx_field_opts;
int efx_composite_fields_ex(efx_ctx* ctx, const efx_field_opts* opts, uint16_t* dst_device);

/* -- PDM audio out (pdm_second_order / write_pcm_16, espflix.ino:73-145) ------------------ */
/* n_streams independent modulators.  pcm: n_streams x n_samples int16 (stream-major, device);
 * state: n_streams x 3 int32 (_i0,_i1,_i2; device, updated in place); dst: n_streams x
 * 2*n_samples uint16 (device).  Asynchronous. */
int efx_pdm(efx_ctx* ctx, int n_streams, const int16_t* pcm_device, int n_samples, int32_t* state_device,
            uint16_t* dst_device);

/* The audio half of MpegDecoder::demux (src/player.cpp:421-433) for a batch of transport streams:
 * the bytes push_audio() (src/video.h:50, src/video.cpp:1006-1019) would receive -- payloads of PID
 * 0x101 / 0x102 behind the PES header, only while the latest audio PES header carried a PTS -- are
 * written to audio_device + i * stride (device memory, stride >= the longest transport stream) and
 * their count to audio_len_device[i].  Feed efx_sbc_decode from there.  ts / len are host pointers;
 * synchronous. */
int efx_demux_audio(efx_ctx* ctx, int n_streams, const uint8_t* const* ts, const size_t* len, uint8_t* audio_device,
                    size_t stride, uint32_t* audio_len_device);

/* -- trick-play index (indexer/indexer.cpp; ESPFlix::idx_hdr, src/espflix.cpp:573-629) -------- */
/* idx_rec as the indexer writes it (indexer/indexer.cpp:22-28): 28 bytes of fields + 4 of padding */
typedef struct efx_idx_rec {
    int64_t first_pts, last_pts;
    uint32_t bin_size, trick_speed, sample_count;
    uint32_t reserved;
} efx_idx_rec;
/* make_index(src) + pts2seq (indexer/indexer.cpp:86-217) for a batch of transport streams, on the
 * device: every video PES that starts a sequence header contributes (PTS, packet number); each
 * bin of bin_size ticks between the first such PTS and the last video PTS gets the packet number
 * of the nearest one.  recs[i] describes stream i (trick_speed[i] is recorded, 1 if the array is
 * NULL); its samples are written to samples + i * samples_cap.  A stream without a sequence header
 * gets sample_count 0.  Host pointers; synchronous. */
int efx_index_streams(efx_ctx* ctx, int n_streams, const uint8_t* const* ts, const size_t* len, const uint32_t* trick_speed,
                      uint32_t bin_size, efx_idx_rec* recs, uint32_t* samples, size_t samples_cap);
/* merge_index (indexer/indexer.cpp:219-237): the bytes of video.idx -- 'IDX', 3, the main /
 * fast-forward / rewind records, then their samples.  Returns the size needed; writes only if it
 * fits in cap. */
size_t efx_idx_build(const efx_idx_rec recs[3], const uint32_t* const samples[3], uint8_t* out, size_t cap);
/* idx_hdr::pts2offset and idx_hdr::pts2pts (src/espflix.cpp:597-627) on the 104-byte header of a
 * video.idx: byte offset of the sample to read (get_index, src/espflix.cpp:823-829) for a
 * main-timeline PTS at speed 0 / 1 / -1, and trick-stream PTS -> main-timeline PTS.  The sample,
 * times 188, is the byte offset at which to start efx_upload_streams(EFX_FORMAT_TS). */
uint32_t efx_idx_pts2offset(const void* idx_hdr, int64_t pts, int speed);
int64_t efx_idx_pts2pts(const void* idx_hdr, int64_t pts, int speed);

/* -- SBC audio decode (sbc_decoder, src/sbc_decoder.cpp:346-378; decode_audio, src/video.cpp:962-989) -- */
/* Bytes of decoder state per stream (the reference's SBC_Decode, src/sbc_decoder.h:12-25).  All
 * zero is sbc_init(). */
size_t efx_sbc_state_bytes(void);
#define EFX_SBC_PROBE_FIRST 1 /* decode frame 0 once more up front and drop that PCM: decode_audio()'s
                                 frame-size probe (src/video.cpp:964-972) synthesises the first frame twice */
/* n_streams independent decoders.  frames: n_frames frames of frame_bytes per stream,
 * stream_stride bytes apart (device); state: n_streams x efx_sbc_state_bytes() (device, updated);
 * pcm: per stream pcm_stride int16 apart (device), frames back to back, blocks x 8 x channels
 * samples each with the channel blocks NOT interleaved (src/sbc_decoder.h:27); ret (device, may
 * be NULL): per (stream, frame) sbc_decoder()'s return value in the low 16 bits (0xFFFF = -1) and
 * its decoded byte count in the high 16; pcm_count (device, may be NULL): samples written per
 * stream.  8 subbands, mono / dual / stereo; joint stereo, 4 subbands, a bad sync byte (and a
 * bitpool above 128, which hangs the reference) are rejected exactly as sbc_decoder() does --
 * including its re-synthesis of the previous subband samples.  Asynchronous.
 * Every stream is decoded in chunks of frames that run side by side (espflix_amd/csrc/k_sbc.hip): what the reference
 * chains from frame to frame -- a rejected frame is synthesised from the samples and under the geometry the state holds
 * -- is resolved into prefix scans over the stream's frames first.  `state` must be zeros or what a call left there
 * (the filter memory is multiplied as 24-bit values, which every value a call can leave fits).
 * frame_bytes x n_frames < 2^28 per call (bit positions inside a stream are 32-bit): longer streams take several
 * calls, the state carries over; EFX_ERR_ARG otherwise. */
int efx_sbc_decode(efx_ctx* ctx, int n_streams, const uint8_t* frames_device, size_t stream_stride, int frame_bytes,
                   int n_frames, void* state_device, int16_t* pcm_device, size_t pcm_stride, uint32_t* ret_device,
                   uint32_t* pcm_count_device, int flags);

/* -- measurement -------------------------------------------------------------------------- */
typedef struct efx_timing {
    float index_ms, parse_ms, recon_ms, total_ms; /* HIP-event stage times, mean over the efx_decode calls
                                                     since efx_set_timing(ctx, 1) (at most the last 64) */
    uint64_t pictures, slices, coefficients, es_bytes;
    float demux_ms;    /* k_demux of the last EFX_FORMAT_TS upload (0 for ES input or timing off at upload) */
    uint32_t timed_calls; /* efx_decode calls averaged in the stage times */
    uint64_t ts_bytes; /* transport-stream bytes of the last upload */
    uint32_t groups;   /* reconstruction groups of the newest call: one k_recon launch per group and picture index; the
                          stage times above are sums over them */
    uint16_t parse_halves; /* parse halves (k_index ... k_parse over a range of streams) of the newest call: they run side
                              by side on the parse streams and feed the reconstruction groups */
    uint16_t mixed;    /* 1: the averaged calls did not all run with the newest call's structure (a call that found the GPU
                          idle is split into groups, one queued behind another is not): per-launch figures derived from
                          the means are off */
    uint32_t recon_launches; /* reconstruction kernel launches of the newest call: groups x pictures with one k_recon launch per
                                picture index, groups with one k_recon_all per group (EFX_OPT_RECON_MODE) */
} efx_timing;
/* Enable HIP-event timing of the decode stages (events recorded on the kernels' own streams);
 * enabling (again) starts a new averaging window. */
int efx_set_timing(efx_ctx* ctx, int enable);
int efx_get_timing(efx_ctx* ctx, efx_timing* out);

/* Launch structure.  Results never depend on it; by default part of it follows what the GPU is doing when a call is
 * queued (a call that finds the reconstruction stream idle is split into groups and its parse kernel is not capped), so the
 * same call sequence can run as different launches from run to run.  Benchmarks and tests pin it. */
typedef enum efx_option {
    EFX_OPT_GROUPS = 1,      /* reconstruction groups per efx_decode: 0 = automatic (above), n >= 1 = always n */
    EFX_OPT_PARSE_CAP = 2,   /* k_parse's residency cap: 0 = only while reconstruction is queued, 1 = always, 2 = never */
    EFX_OPT_RECON_MODE = 3,  /* 0 = one k_recon launch per picture index (default); 1 or 2 = ONE launch per group for all picture
                                indices (k_recon_all: a stream's pictures ordered by a per-stream counter -- bit-identical, on a
                                par one call at a time, 8 % slower back to back: profiles/r5_recon_all.md) */
    EFX_OPT_RECON_WAVES = 4, /* k_recon_all with EFX_OPT_RECON_ITEMS = 0: workgroups (waves) per compute unit; 0 = default (18) */
    EFX_OPT_RECON_ITEMS = 6, /* k_recon_all: items -- (picture, stream, 64 blocks) -- a wave takes before it ends and frees its
                                slot (default 16); 0 = as many as there are (a grid of what the chip holds) */
    EFX_OPT_SBC_SERIAL = 7,  /* 1 = efx_sbc_decode runs every stream through the one-wave-per-stream kernel (the comparison the tests
                                run the frame-parallel kernels against); 0 = default */
    EFX_OPT_DEMUX_FUSED = 8, /* 1 = transport-stream uploads run the ONE-PASS demultiplexer (a ticket per chunk, decoupled look-back over
                                chunk descriptors, one launch: k_demux_fused) instead of scan / prefix / gather.  Bit-exact and tested,
                                but measured slower (82 against 63 us per 48.8 MB: profiles/r6_demux.md) -- 0 = default */
    EFX_OPT_RECON_SPINS = 5  /* read only: polls the reconstruction waves of the most recent call spent waiting for a predecessor
                                picture (synchronises) */
} efx_option;
/* Debugging aid of the guard-page allocator (EFX_GUARD=1 / 2 in the environment: every device buffer of the library and of
 * efx_device_alloc becomes its own mapping that ends -- mode 2: starts -- on an unmapped page, so that a kernel reading or
 * writing past a buffer faults at once): writes one word at `offset_bytes` from the start of a buffer (negative: in front of
 * it).  tools/guard_selftest.py uses it to show that the allocator catches what it is there to catch. */
int efx_debug_poke(efx_ctx* ctx, void* dptr, size_t bytes, long long offset_bytes);
/* Debugging aid: word 0 of each of the 64 header lines of k_recon_all's hand-over words after the most recent call (queue
 * heads 0-7, spins 8, abort 9; 16 ... per-phase wave times in a -DEFX_RA_STATS build). */
int efx_debug_recon_stats(efx_ctx* ctx, uint32_t out[64]);
int efx_set_option(efx_ctx* ctx, int option, int value);
int efx_get_option(efx_ctx* ctx, int option, int* value);

/* -- several devices of one node ----------------------------------------------------------------- */
/* The reference decodes one stream on one core; its batch form here shards by STREAM and nothing else (SURVEY.md 8e,
 * BASELINE.json north_star): streams are independent, so stream k of a batch of n goes to device floor(k * R / n) of
 * the R devices -- contiguous blocks -- and no collective touches the data path.  An efx_multi owns one efx_ctx and one
 * host thread per device (a HIP context is bound to the thread that drives it); every call below fans out to the
 * devices, each working on its own block, and returns when all of them have QUEUED (decode) or FINISHED (upload,
 * sync, queries) their part.  The per-device contexts are the ordinary ones: efx_multi_context(m, r) hands out device
 * r's for everything this section does not wrap (composite fields, PDM, timing ...), to be used from the caller's
 * thread only while no efx_multi_* call is in flight.  What MpegDecoder::run() is to one stream
 * (src/player.cpp:1355-1367) efx_multi_decode is to a node's worth of them. */
/* NUMA placement of a device's host side (host only, no device call): the node sysfs publishes for a PCI device
 * ("0000:c1:00.0", as hipDeviceGetPCIBusId prints it; -1 = unknown / single node), that node's CPUs, and binding the
 * CALLING thread to them (returns how many CPUs the new mask holds, 0 = left alone: unknown node, or none of its CPUs are
 * available to this process).  efx_multi_create binds each device's worker thread this way before the thread creates its
 * context -- its pinned staging buffers are then first touched on the device's own node -- unless EFX_NUMA=0.  A
 * one-process-per-GPU launcher calls efx_numa_bind_thread itself before efx_create (bench.py does, per rank). */
int efx_numa_node_of_pci(const char* pci_bus_id);
int efx_numa_cpus_of_node(int node, int* cpus, int cap);
int efx_numa_bind_thread(int node);

typedef struct efx_multi efx_multi;
/* first stream of part `part` when `total` streams are dealt to `parts` devices: ceil(part * total / parts); part ==
 * parts gives total.  Pure function (no device needed): the partition the multi-device calls and bench.py use. */
int efx_partition_first(int total, int parts, int part);
/* cfg->max_streams is the PER-DEVICE capacity, cfg->device is ignored; devices[r] = HIP ordinal of device r. */
int efx_multi_create(const efx_config* cfg, const int* devices, int n_devices, efx_multi** out);
void efx_multi_destroy(efx_multi* m);
int efx_multi_device_count(const efx_multi* m);
efx_ctx* efx_multi_context(efx_multi* m, int r);
const char* efx_multi_last_error(const efx_multi* m);
/* efx_upload_streams for a batch of n_streams <= n_devices x max_streams streams, dealt as above */
int efx_multi_upload_streams(efx_multi* m, int n_streams, const uint8_t* const* data, const size_t* len, int format);
/* where stream k of the last upload lives: device index and its index in that device's batch */
int efx_multi_locate(const efx_multi* m, int stream, int* device_index, int* local_stream);
int efx_multi_decode(efx_multi* m);  /* efx_decode on every device; asynchronous */
int efx_multi_sync(efx_multi* m);
int efx_multi_reset(efx_multi* m);
/* per stream of the last upload, in batch order: pictures decoded, status bits (either pointer may be NULL) */
int efx_multi_results(efx_multi* m, int* n_pictures, uint32_t* status);
/* efx_frame_hashes of every stream of the last upload, in batch order: out[n_streams][ring_depth] */
int efx_multi_frame_hashes(efx_multi* m, uint64_t* out);

/* raw device allocations for callers without their own allocator (bench, tests) */
int efx_device_alloc(efx_ctx* ctx, size_t bytes, void** dptr);
int efx_device_free(efx_ctx* ctx, void* dptr);
int efx_memcpy_h2d(efx_ctx* ctx, void* dst_device, const void* src, size_t bytes);
int efx_memcpy_d2h(efx_ctx* ctx, void* dst, const void* src_device, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* EFX_H */
