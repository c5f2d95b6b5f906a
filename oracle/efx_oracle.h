/* efx_oracle.h -- CPU restatement of the espflix hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the HIP implementation in espflix_amd/csrc.  It is a
 * plain-C restatement of the reference algorithms, each function citing the reference
 * file:line it follows.  It is pinned against the real reference (compiled unmodified into
 * oracle/_ref/ by oracle/Makefile) on the two embedded clips (video and audio), on synthetic
 * streams and hostile muxing, on composite fields incl. overlay / slide, on PDM words, on SBC
 * frames and tables, and on the indexer's video.idx -- see tests/test_oracle_vs_ref.py and
 * tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (espflix_amd/) never links, imports or executes anything in oracle/.
 */
#ifndef EFX_ORACLE_H
#define EFX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFXO_FRAME_BYTES 101376           /* 12 strips x 16 rows x 528 B (video.h:30-34) */

enum { EFXO_FMT_ES = 0, EFXO_FMT_TS = 1 };

/* Decode a whole elementary stream (EFXO_FMT_ES) or transport stream (EFXO_FMT_TS) exactly as
 * MpegDecoder::run() would (player.cpp:1355-1367) and record every frame handed to
 * push_video() (player.cpp:692-702).
 *   flush_last  != 0: also push the final picture (flush_picture(1), as load_poster does,
 *                     espflix.cpp:1068).
 *   frames_out  NULL or room for max_frames x EFXO_FRAME_BYTES
 *   pts_out     NULL or max_frames int64 (pts handed to push_video)
 *   hash_out    NULL or max_frames uint64 (FNV-1a-64 of the 101376 frame bytes)
 * In ES mode there is no PES layer, so picture i is given pts = i (a pts is "seen" before the
 * first picture, i.e. every picture start pushes/swaps).
 * Returns the number of frames pushed (may exceed max_frames; extra frames are only counted),
 * or a negative number on argument error. */
long efxo_decode(const uint8_t* data, size_t len, int format, int flush_last,
                 uint8_t* frames_out, int64_t* pts_out, uint64_t* hash_out, long max_frames);

/* Parse trace for tests/parse_harness.cpp: when a callback is set, efxo_decode reports, in decode order,
 *   EFXO_T_SLICE  a = picture index, b = slice start code, c = picture_coding_type | full_pel << 4 | r_size << 8 |
 *                 (slice decoded ? 1 << 16 : 0), e = loaded quantiser matrices in force
 *   EFXO_T_MB     a = macroblock address, b = intra | skipped << 1 | quantiser_scale << 2, c / e = half-pel luma vector
 *   EFXO_T_COEF   a = block, b = scan position, c = signed level before dequantisation (intra DC: the DC value)
 *   EFXO_T_BLOCK  a = block, b = 0 decoded / -1 abandoned (ran past 64) / -2 invalid code, c = entries so far
 *   EFXO_T_ESCAPE (before the EFXO_T_COEF it belongs to) a = form of the level, player.cpp:1092-1099: 0 one byte,
 *                 1 "00 xx" = xx, 2 "80 xx" = xx - 256; b = run, c = level
 *   EFXO_T_SLICE_EXTRA (after the EFXO_T_SLICE it belongs to, only when the header has any) a = number of
 *                 extra_information_slice bytes skipped (player.cpp:1261-1262), b = the last one, c = picture type */
enum { EFXO_T_SLICE = 0, EFXO_T_MB = 1, EFXO_T_COEF = 2, EFXO_T_BLOCK = 3, EFXO_T_ESCAPE = 4, EFXO_T_SLICE_EXTRA = 5, EFXO_T_SLICE_AT = 6 };
typedef void (*efxo_trace_fn)(void* user, int kind, int a, int b, int c, int e);
void efxo_set_trace(efxo_trace_fn fn, void* user);

/* TS -> video ES (PID 0x100 payloads, PES headers stripped; player.cpp:381-493).  Writes at
 * most es_cap bytes, returns the ES length.  pic_pts_out/max_pics (optional) receive the PES
 * pts latched by flush_picture for each picture start code in order of appearance. */
size_t efxo_ts_to_es(const uint8_t* ts, size_t len, uint8_t* es_out, size_t es_cap);

/* FNV-1a-64 (offset basis cbf29ce484222325, prime 100000001b3). */
uint64_t efxo_fnv1a64(const uint8_t* p, size_t n, uint64_t h);

/* Composite video: restatement of video_init + video_isr (video.cpp:572-630,1122-1198).
 * frames2 = Frame[0] then Frame[1] (2 x EFXO_FRAME_BYTES), front frame = 0, overlay off, no
 * h-scroll.  Produces nfields fields starting with _frame_counter = frame_counter0, each
 * line_count x line_width u16.  Returns samples written per field, or <0. */
long efxo_video_field(const uint8_t* frames2, int ntsc, int frame_counter0, int nfields, uint16_t* out);
/* The same with the displayed frame `front` (0/1), a per-field _hscroll value (NULL = 0; the
 * two-frame slide of video.cpp:1146-1154) and the 80 x 16 overlay + progress bar of composite()
 * (video.cpp:845-887): overlay = 1280 bytes or NULL, blend = _video_composite_blend at the first
 * field (decremented per field while > 0, video.cpp:1192-1193), progress =
 * _video_composite_progress. */
long efxo_video_field_ex(const uint8_t* frames2, int ntsc, int frame_counter0, int nfields, int front,
                         const int16_t* hscroll, const uint8_t* overlay, int blend, int progress, uint16_t* out);
/* geometry: {line_width,line_count,hsync,hsync_long,hsync_short,burst_start,burst_width,active_start} */
void efxo_video_params(int ntsc, int32_t out8[8]);
/* the 768-entry colour LUT video_init leaves in _color_tab (video.cpp:584-591) */
void efxo_color_tab(int ntsc, uint32_t out768[768]);

/* zig-zag scan and IDCT pre-multiplier tables as the restatement derives them */
void efxo_tables(uint8_t zz_out[64], uint8_t premul_out[64]);

/* PDM: restatement of pdm_second_order (espflix.ino:73-107).  state = {_i0,_i1,_i2}. */
void efxo_pdm_second_order(int32_t state[3], uint16_t* dst, const int16_t* src, int len);
/* write_pcm_16 (espflix.ino:123-145): s==NULL -> 256 x 0xAAAA; *beep>0 -> sine burst. */
void efxo_write_pcm_16(int32_t state[3], int* beep, const int16_t* s, int n, uint16_t out256[256]);

/* SBC audio: restatement of sbc_decoder.cpp:74-373 (8 subbands, mono / dual / stereo; joint
 * stereo and 4 subbands are rejected as in the reference).  The state is the reference's
 * SBC_Decode (sbc_decoder.h:12-25); zero it with efxo_sbc_init. */
typedef struct efxo_sbc {
    uint8_t inited, frequency, blocks, channels, mode, allocation, subbands, bitpool;
    int32_t sb_sample[16][2][8];
    int32_t v[2][160 + 10];
    uint8_t v_offset[2][16];
} efxo_sbc;
void efxo_sbc_init(efxo_sbc* s);
/* the synthesis matrix SBC_syn_8[128] and window SBC_proto_8[80] (sbc_decoder.cpp:41-71) */
void efxo_sbc_tables(int32_t syn128[128], int32_t proto80[80]);
/* sbc_decoder(): returns the frame length consumed (or -1), writes blocks*8*channels int16
 * (channel blocks NOT interleaved) to dst and their byte count to *decoded. */
int efxo_sbc_decode(efxo_sbc* s, const uint8_t* src, int src_len, int16_t* dst, int* decoded);
/* The bytes push_audio() receives for a transport stream (MpegDecoder::demux audio branch,
 * player.cpp:421-433): payloads of PID 0x101 / 0x102 behind the PES header, only while the
 * latest audio PES header carried a PTS.  Returns the byte count (may exceed cap). */
size_t efxo_ts_audio_es(const uint8_t* ts, size_t len, uint8_t* out, size_t cap);

/* Trick-play index: restatement of the indexer (indexer/indexer.cpp:22-36,86-237) and of the
 * player's idx_hdr arithmetic (src/espflix.cpp:573-629).  video.idx = idx_hdr (104 bytes: sig
 * 'IDX', len 3, three idx_rec of 32 bytes) followed by the uint32 samples of the main, fast-forward
 * and rewind streams; a sample is the 188-byte packet number of the PES that starts the sequence
 * header nearest (in PTS) to its bin. */
typedef struct efxo_idx_rec {
    int64_t first_pts, last_pts;
    uint32_t bin_size, trick_speed, sample_count;
    uint32_t pad;
} efxo_idx_rec;
/* make_index(src, idxs), indexer.cpp:86-176: sequence-start (pts, packet) pairs of one transport
 * stream; returns their number (may exceed cap), first_pts = origin, last_pts = last video PES pts */
long efxo_ts_sequences(const uint8_t* ts, size_t len, int64_t* first_pts, int64_t* last_pts, int64_t* seq_pts,
                       uint32_t* seq_pos, long cap);
/* make_index(path) + merge_index, indexer.cpp:180-237,301-308: the bytes of video.idx for the
 * three streams (main, fwd, rwd); returns the size (may exceed cap), 0 if a stream has no sequence */
size_t efxo_make_idx(const uint8_t* const ts[3], const size_t len[3], uint8_t* out, size_t cap);
/* idx_hdr::pts2offset / pts2pts, espflix.cpp:589-627 (hdr = the first 104 bytes of video.idx) */
uint32_t efxo_idx_pts2offset(const uint8_t* hdr, int64_t pts, int speed);
int64_t efxo_idx_pts2pts(const uint8_t* hdr, int64_t pts, int speed);

#ifdef __cplusplus
}
#endif
#endif
