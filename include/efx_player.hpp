// efx_player.hpp -- source-level drop-in for the espflix player surface, backed by libefx.
//
// The reference host player (src/espflix.cpp) is written against a handful of C++ names:
//
//   class Frame                        src/video.h:36-44
//   class Buffer                       src/streamer.h:139-143
//   class MpegDecoder                  src/player.h:34-165   (public part: ctor, push_full,
//                                      pop_empty, reset, run, get_pts, flush_picture)
//   void push_video(Frame*,int,int64_t,int)       src/video.h:49   (up-call, defined by the host)
//   void video_init(int ntsc) / video_reset() / video_pause(int)   src/video.h:46-48
//   extern "C" void video_isr(volatile void* buf)                  src/video.cpp:47,1122
//   void write_pcm_16(const int16_t*, int, int)                    espflix.ino:123
//   DECODER_RUN / DECODER_PAUSED event bits, set/clear/wait/get_events   src/streamer.h:34-40,119-123
//
// This header re-declares exactly that surface (same names, argument meaning, threading and
// error behaviour) for batch = 1 on top of the C-ABI in efx.h, so player code written against
// the reference compiles against it unchanged and gets its pictures from the MI355X path.
// Include it in ONE translation unit with EFX_PLAYER_IMPLEMENTATION defined to get the bodies.
//
// Behavioural notes (all visible through the reference API only):
//   * MpegDecoder::run() lives on its own thread, blocks in pop of the full queue, and hands each
//     picture to push_video() on that thread, in order, with the PTS the reference would latch;
//     like the reference it does NOT push the last picture of a stream until flush_picture(mode)
//     is called, pushes nothing before a picture has latched a PES PTS, and parks in pause()
//     (DECODER_PAUSED) at end of stream.
//   * STREAMING: Buffers are decoded while they arrive, in windows cut where a video PES starts with
//     a sequence, group or picture start code (every picture of the reference's own files and of
//     ffmpeg's muxer starts a PES): a window is decoded when a sequence header opens the next one,
//     when it holds kWindowPictures pictures, at the zero-length Buffer -- and whenever no further Buffer is
//     waiting (the feeder is not ahead of the decoder: a real-time play), so that a picture is pushed when the
//     PES of its successor has arrived, as the reference pushes it at its successor's header.  Only a feeder
//     faster than the decoder makes pictures arrive in batches (up to one window late); a play may be of any length.
//     The decoder state travels from window to window on the device (frame index, PTS latch, reference
//     frame: efx.h, efx_decode), exactly as the reference's does from Buffer to Buffer.  A stream whose
//     PES packets never start at a picture is decoded at its end (one window of at most
//     kWindowBytes; what does not fit is reported on stderr, never dropped silently).
//   * Audio: transport packets of PID 0x101 / 0x102 are handed to push_audio() as MpegDecoder::demux does
//     (player.cpp:421-433), from the decoder thread.
//   * Frame strips are host memory (the UI draws into them, src/espflix.cpp:62-84); decoded
//     pictures are copied into them before push_video(), the two Frames are uploaded to the device ring
//     at the start of a play (a play that opens with P pictures predicts from what the Frames hold), and
//     video_isr() re-uploads the front Frame once per field, so host-side drawing is honoured.
//
// Platform layer.  Defined on its own this header also declares the few platform names the player
// code needs (Q, Buffer, the event word); compiled INTO the reference tree in place of player.h /
// video.h (include/espflix_dropin/) it takes them from the reference's own streamer.h
// (EFX_PLAYER_USE_REFERENCE_PLATFORM), so that the unmodified espflix.cpp and streamer.cpp build
// against it.
#ifndef EFX_PLAYER_HPP
#define EFX_PLAYER_HPP

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <chrono>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "efx.h"

#define FB_WIDTH 352
#define FB_HEIGHT 192
#define FB_STRIDE (FB_WIDTH * 3 / 2)
#define FB_SLICE_HEIGHT 16
#define FB_SLICES (FB_HEIGHT / FB_SLICE_HEIGHT)

#ifdef EFX_PLAYER_USE_REFERENCE_PLATFORM
#include "streamer.h"  // the reference's: Q, Buffer, DECODER_* events, start_thread, printf -> printf_nano
#else
enum { DECODER_RUN = 2, DECODER_PAUSED = 4, AUDIO_READY = 8, VIDEO_READY = 16, DNS_READY = 256 };

int get_events();
void clear_events(int i);
void wait_events(int i);
void set_events(int i);

class Q {
    std::queue<const void*> queue;
    mutable std::mutex guard;
    std::condition_variable signal;

  public:
    void push(const void* data);
    const void* pop();
    bool empty();
    int waiting();
};

class Buffer {
  public:
    uint32_t len;
    uint8_t data[8 * 188];
};
#endif

class Frame {
  public:
    uint8_t* _slices[FB_SLICES];
    void init();
    uint8_t* get_y(int y);
    uint8_t* get_cr(int y);
    uint8_t* get_cb(int y);
    void erase();
};

// up-calls the host provides, exactly as in the reference
void push_video(Frame* f, int front, int64_t pts, int mode);
void push_audio(const uint8_t* data, int len, int64_t pts, bool pes_complete);

class MpegDecoder {
  public:
    Frame* _fb[2];
    int _fb_index;
    int64_t _pts;
    int64_t _last_pts;

    MpegDecoder(Frame* fb0, Frame* fb1);
    ~MpegDecoder();
    void push_full(Buffer* b);  // from the feeder thread
    Buffer* pop_empty();
    void reset();
    void run();  // never returns
    int64_t get_pts();
    void flush_picture(int mode = 0);

    enum { kWindowPictures = 30, kMaxPictures = 64, kWindowBytes = 32 << 20 };

  protected:
    void pause();
    void feed(const uint8_t* packets, size_t bytes);  // transport packets of one Buffer
    void decode_window(size_t bytes, bool next_started = false);  // decode the first `bytes` of the window, keep the rest
    void seed_ring();
    Q _empty_q;
    Q _full_q;
    std::vector<uint8_t> _win;  // transport packets not yet decoded
    size_t _cut;                // newest cut candidate inside _win (0 = none)
    int _win_pictures;          // picture-starting PES packets in _win
    bool _play_started;         // the ring has been seeded for this play
    efx_ctx* _ctx;
    bool _have_last;            // a decoded picture is waiting for the next flush_picture()
    // audio branch of demux (player.cpp:421-433)
    int64_t _audio_pts;
    int _audio_expected, _audio_mark;
    std::vector<uint8_t> _frame;
};

void video_init(int ntsc);
void video_reset();
void video_pause(int p);
extern "C" void video_isr(volatile void* buf);
// publish the frames the line callback displays (the reference's push_video does this through
// file-scope globals _frames/_next_frame, src/video.cpp:1027,1054)
// `mode` is push_video's: 2 / 3 start the ease-in / ease-out slide between the two Frames at the
// flip (src/video.cpp:1047-1050,1166-1171).
void efx_video_present(Frame* frames, int front, int mode = 0);
int efx_video_line_width();
int efx_video_line_count();

// the time / progress-bar overlay the host draws into (src/video.h:52-55, src/video.cpp:838-843)
#define VIDEO_COMPOSITE_WIDTH 80
#define VIDEO_COMPOSITE_HEIGHT 16
#define VIDEO_COMPOSITE_PROGRESS_WIDTH (352 - VIDEO_COMPOSITE_WIDTH - 32)
extern uint8_t _video_composite[VIDEO_COMPOSITE_HEIGHT * VIDEO_COMPOSITE_WIDTH];
extern int _video_composite_blend;     // -1 always show, 1-31 blend, >=32 show
extern int _video_composite_progress;

void write_pcm_16(const int16_t* s, int n, int channels);
void beep();
// where write_pcm_16 delivers its 256 PDM words (the reference calls i2s_write, espflix.ino:142)
typedef void (*efx_pdm_sink)(const uint16_t* words, int n);
void efx_set_pdm_sink(efx_pdm_sink sink);

#ifdef EFX_PLAYER_IMPLEMENTATION
// =================================================================================================

#ifndef EFX_PLAYER_USE_REFERENCE_PLATFORM
namespace efx_player_detail {
inline std::mutex& ev_guard()
{
    static std::mutex m;
    return m;
}
inline std::condition_variable& ev_signal()
{
    static std::condition_variable c;
    return c;
}
inline int& ev_word()
{
    static int w = 0;
    return w;
}
}  // namespace efx_player_detail

int get_events()
{
    std::lock_guard<std::mutex> l(efx_player_detail::ev_guard());
    return efx_player_detail::ev_word();
}
void clear_events(int i)
{
    std::lock_guard<std::mutex> l(efx_player_detail::ev_guard());
    efx_player_detail::ev_word() &= ~i;
    efx_player_detail::ev_signal().notify_all();
}
void set_events(int i)
{
    std::lock_guard<std::mutex> l(efx_player_detail::ev_guard());
    efx_player_detail::ev_word() |= i;
    efx_player_detail::ev_signal().notify_all();
}
void wait_events(int i)
{
    std::unique_lock<std::mutex> l(efx_player_detail::ev_guard());
    while (!(efx_player_detail::ev_word() & i))
        efx_player_detail::ev_signal().wait(l);
}

void Q::push(const void* data)
{
    {
        std::lock_guard<std::mutex> lock(guard);
        queue.push(data);
    }
    signal.notify_one();
}
const void* Q::pop()
{
    std::unique_lock<std::mutex> lock(guard);
    while (queue.empty())
        signal.wait(lock);
    const void* v = queue.front();
    queue.pop();
    return v;
}
bool Q::empty()
{
    std::unique_lock<std::mutex> lock(guard);
    return queue.empty();
}
int Q::waiting()
{
    std::unique_lock<std::mutex> lock(guard);
    return (int)queue.size();
}

#endif  // EFX_PLAYER_USE_REFERENCE_PLATFORM

void Frame::init()
{
    for (int i = 0; i < FB_SLICES; i++) {
        _slices[i] = (uint8_t*)malloc(FB_STRIDE * FB_SLICE_HEIGHT + 4);
        memset(_slices[i], 0, FB_STRIDE * FB_SLICE_HEIGHT + 4);
    }
}
uint8_t* Frame::get_y(int y) { return _slices[y >> 4] + (y & 0xF) * FB_STRIDE; }
uint8_t* Frame::get_cr(int y) { return _slices[y >> 3] + (y & 0x7) * FB_STRIDE + FB_WIDTH; }
uint8_t* Frame::get_cb(int y) { return _slices[y >> 3] + ((y & 0x7) + 8) * FB_STRIDE + FB_WIDTH; }
void Frame::erase()
{
    for (int i = 0; i < FB_SLICES; i++)
        memset(_slices[i], 0x30, FB_STRIDE * FB_SLICE_HEIGHT + 4);
}

MpegDecoder::MpegDecoder(Frame* fb0, Frame* fb1)
    : _cut(0), _win_pictures(0), _play_started(false), _ctx(0), _have_last(false), _audio_pts(-1), _audio_expected(0), _audio_mark(0),
      _frame(EFX_FRAME_BYTES)
{
    _fb[0] = fb0;
    _fb[1] = fb1;
    _fb_index = 1;  // the reference starts with _reference = fb0, _current = fb1 (player.cpp:358-360)
    _last_pts = _pts = -1;
    for (int i = 0; i < 4; i++)
        _empty_q.push(new Buffer());
    efx_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_streams = 1;
    cfg.max_pictures = kMaxPictures;
    cfg.ring_depth = kMaxPictures + 1;  // every picture of a decode pass stays until it has been handed out
    cfg.max_stream_bytes = kWindowBytes;
    if (efx_create(&cfg, &_ctx) != EFX_OK) {
        fprintf(stderr, "MpegDecoder: efx_create failed (a gfx950 device is required)\n");
        abort();
    }
    // one pass over an empty transport stream now: the transport-stream buffers, the code objects and the queues
    // exist before the first picture arrives (the first window of a real-time play is not the slow one)
    uint8_t null_packet[188];
    memset(null_packet, 0xFF, sizeof(null_packet));
    null_packet[0] = 0x47;
    null_packet[1] = 0x1F;
    null_packet[3] = 0x10;
    const uint8_t* ptr = null_packet;
    size_t len = sizeof(null_packet);
    if (efx_upload_streams(_ctx, 1, &ptr, &len, EFX_FORMAT_TS) == EFX_OK && efx_decode(_ctx) == EFX_OK)
        efx_sync(_ctx);
}

MpegDecoder::~MpegDecoder() { efx_destroy(_ctx); }

void MpegDecoder::push_full(Buffer* b) { _full_q.push(b); }
Buffer* MpegDecoder::pop_empty() { return (Buffer*)_empty_q.pop(); }

void MpegDecoder::reset()  // player.cpp:439-453: called by the feeder while the decoder is paused
{
    while (!_full_q.empty())
        _empty_q.push(_full_q.pop());
    _win.clear();
    _cut = 0;
    _win_pictures = 0;
    _play_started = false;
    video_reset();
    _last_pts = -1;
    _audio_pts = -1;
    _have_last = false;
    efx_play_reset(_ctx);  // _last_pts = -1 on the device too; frame index and newest PTS survive, as in the reference
}

int64_t MpegDecoder::get_pts() { return _last_pts; }

void MpegDecoder::flush_picture(int mode)  // player.cpp:692-702
{
    if (_have_last && (_last_pts != -1 || mode)) {
        push_video(_fb[0], _fb_index & 1, _last_pts, mode);
        _fb_index++;
        _have_last = false;
    }
    if (!mode)
        _last_pts = _pts;
}

void MpegDecoder::pause()  // player.cpp:1342-1352
{
    clear_events(DECODER_RUN);
    set_events(DECODER_PAUSED);
    if (_empty_q.empty())
        _empty_q.push(0);  // unstick a feeder waiting in pop_empty()
    wait_events(DECODER_RUN);
    clear_events(DECODER_PAUSED);
}

// The two host Frames into the device ring, at the slots the next picture and its reference occupy
// (_current / _reference): the reference decodes INTO the Frames the UI draws on, so a play that opens
// with P pictures predicts from whatever they hold.
void MpegDecoder::seed_ring()
{
    uint32_t fi = 1;
    if (efx_stream_state(_ctx, 0, &fi, 0, 0) != EFX_OK)
        return;
    const uint32_t depth = kMaxPictures + 1;
    for (int k = 0; k < 2; k++) {  // k = 0: _current (slot fi), k = 1: _reference (slot fi - 1)
        const Frame* f = _fb[(_fb_index - k) & 1];
        for (int s = 0; s < FB_SLICES; s++)
            memcpy(&_frame[(size_t)s * EFX_STRIP_BYTES], f->_slices[s], EFX_STRIP_BYTES);
        efx_upload_frame(_ctx, 0, (int)((fi - (uint32_t)k) % depth), &_frame[0]);
    }
}

void MpegDecoder::decode_window(size_t bytes, bool next_started)
{
    if (bytes > _win.size())
        bytes = _win.size();
    if (!_play_started) {
        seed_ring();
        _play_started = true;
    }
    const uint8_t* ptr = _win.empty() ? (const uint8_t*)"" : &_win[0];
    size_t len = bytes;
    if (len > (size_t)kWindowBytes) {
        fprintf(stderr, "MpegDecoder: %zu bytes without a picture-aligned PES; only the first %d are decoded\n", len, (int)kWindowBytes);
        len = kWindowBytes / 188 * 188;
    }
    if (efx_upload_streams(_ctx, 1, &ptr, &len, EFX_FORMAT_TS) != EFX_OK) {
        fprintf(stderr, "MpegDecoder: %s\n", efx_last_error(_ctx));
        len = 0;
    }
    // passes over the window: the first for as many pictures as picture-aligned PES packets were counted (a real-time
    // play: one or two -- that many reconstruction launches, not kMaxPictures of them), more passes if more turn up
    int budget = _win_pictures + 1 < (int)kMaxPictures ? _win_pictures + 1 : (int)kMaxPictures;
    for (int first = 0; len;) {
        if (efx_decode_range(_ctx, first, budget) != EFX_OK || efx_sync(_ctx) != EFX_OK) {
            fprintf(stderr, "MpegDecoder: %s\n", efx_last_error(_ctx));
            break;
        }
        int n = 0;
        uint32_t status = 0;
        efx_picture_count(_ctx, 0, &n);
        efx_stream_status(_ctx, 0, &status);
        if (status & ~(uint32_t)EFX_STREAM_TRUNCATED)
            fprintf(stderr, "MpegDecoder: stream status %u (see efx.h)\n", status);
        for (int i = 0; i < n; i++) {
            int64_t pts = -1;
            int slot = 0;
            efx_picture_pts(_ctx, 0, i, &pts);
            _pts = pts;        // what picture() finds latched (player.cpp:417-418,700)
            flush_picture(0);  // picture start: push the previous picture, swap buffers (player.cpp:704-706)
            Frame* dst = _fb[_fb_index & 1];
            efx_stream_picture_slot(_ctx, 0, i, &slot);
            efx_download_frame(_ctx, 0, slot, &_frame[0]);
            for (int s = 0; s < FB_SLICES; s++)
                memcpy(dst->_slices[s], &_frame[(size_t)s * EFX_STRIP_BYTES], EFX_STRIP_BYTES);
            _have_last = true;
        }
        if (!(status & EFX_STREAM_TRUNCATED) || n <= 0)
            break;
        first += n;
        budget = kMaxPictures;
    }
    _win.erase(_win.begin(), _win.begin() + (ptrdiff_t)bytes);
    _cut = 0;
    _win_pictures = 0;
    // The window was cut where the PES of the NEXT picture starts: the reference, reaching that picture's header, pushes
    // the one before it now (flush_picture, player.cpp:692-702) -- so does this.  (The PTS the next picture latches is set
    // when it is decoded; the push itself only needs the one already latched.)
    if (next_started && _have_last && _last_pts != -1) {
        push_video(_fb[0], _fb_index & 1, _last_pts, 0);
        _fb_index++;
        _have_last = false;
    }
}

namespace efx_player_detail {
inline int64_t parse_pts(const uint8_t* d, int flags)  // player.cpp:299-307
{
    flags = (flags >> 2) & 0x30;
    if ((d[0] & 0xF0) != flags)
        return -1;
    int64_t n = ((int64_t)(d[0] & 0x0E)) << 29;
    n |= (int64_t)d[1] << 22;
    n |= (int64_t)(d[2] & 0xFE) << 14;
    n |= (int64_t)d[3] << 7;
    n |= d[4] >> 1;
    return n;
}
}  // namespace efx_player_detail

// One Buffer of transport packets: audio goes to push_audio() at once (player.cpp:421-433), video
// accumulates in the window, which is decoded at cut points (see the header of this file).
void MpegDecoder::feed(const uint8_t* pk, size_t bytes)
{
    for (size_t o = 0; o + 188 <= bytes; o += 188) {
        const uint8_t* d = pk + o;
        if (d[0] != 0x47) {  // "ts lost sync": the reference feeds a zero byte and drops the rest of the Buffer (player.cpp:476-480)
            _win.insert(_win.end(), d, d + 188);
            continue;
        }
        const int pid = ((d[1] << 8) + d[2]) & 0x1FFF;
        const bool pusi = d[1] & 0x40;
        const uint8_t* data = d + 4;
        if (d[3] & 0x20)
            data = d + 5 + d[4];
        const uint8_t* end = d + 188;
        const bool has_payload = d[3] & 0x10;
        if (pid == 0x100) {
            if (has_payload && pusi && data + 13 <= end) {
                const uint8_t* es = data + 9 + data[8];  // PES payload (player.cpp:391-393)
                if (es + 4 <= end && es[0] == 0 && es[1] == 0 && es[2] == 1 && (es[3] == 0xB3 || es[3] == 0xB8 || es[3] == 0x00)) {
                    // a picture (group, sequence) starts a PES here: everything before this packet is whole pictures
                    const size_t at = _win.size();
                    if (at && (es[3] == 0xB3 || _win_pictures >= kWindowPictures || at + 188 > (size_t)kWindowBytes)) {
                        decode_window(at, true);
                    } else if (at)
                        _cut = at;
                    _win_pictures++;
                }
            }
            if (_win.size() + 188 > (size_t)kWindowBytes && _cut) {
                decode_window(_cut, true);
            }
            _win.insert(_win.end(), d, d + 188);
        } else if ((pid == 0x101 || pid == 0x102) && has_payload) {
            int64_t pts = -1;
            int expected = 0;
            const uint8_t* payload = data;
            if (pusi && data + 14 <= end) {
                expected = (data[4] << 8) | data[5];
                const int flags = (data[6] << 8) | data[7];
                payload = data + 9 + data[8];
                if (expected)
                    expected -= 3 + data[8];
                if (flags & 0x0080)
                    pts = efx_player_detail::parse_pts(data + 9, flags);
            }
            if (pusi) {
                _audio_expected = expected;
                _audio_mark = 0;
                _audio_pts = pts;
            }
            if (_audio_pts != -1 && payload <= end) {
                _audio_mark += (int)(end - payload);
                push_audio(payload, (int)(end - payload), pts, _audio_mark == _audio_expected);
            }
        }
    }
}

void MpegDecoder::run()  // player.cpp:1355-1367
{
    for (;;) {
        if (!(get_events() & DECODER_RUN))
            pause();
        Buffer* b = (Buffer*)_full_q.pop();
        if (!b)
            continue;
        const bool eos = (int32_t)b->len <= 0 || b->len > sizeof(b->data);
        if (!eos)
            feed(b->data, b->len);
        _empty_q.push(b);
        // Adaptive window: no Buffer is waiting, so the feeder is not ahead of the decoder (a real-time play): the whole
        // pictures the window holds are decoded now and a picture is pushed as soon as the PES of its successor has
        // arrived -- the reference's own latency.  While Buffers queue up (a feeder faster than the decoder) pictures are
        // batched, up to kWindowPictures per pass over the GPU.
        // (a feeder that is merely between two push_full calls refills the queue within microseconds: give it 300)
        if (!eos && _cut && _full_q.empty()) {
            for (int spin = 0; spin < 6 && _full_q.empty(); spin++)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (_full_q.empty())
                decode_window(_cut, true);
        }
        if (eos) {  // zero-length Buffer: the decoder pads with a sequence_end code and pauses (player.cpp:469-473,1324-1327)
            decode_window(_win.size());
            pause();
        }
    }
}

// ---- video out ---------------------------------------------------------------------------------

namespace efx_player_detail {
struct VideoState {
    efx_ctx* ctx;
    int ntsc;
    efx_video_params vp;
    Frame* frames;
    int front;
    int line_counter, frame_counter;
    int hscroll, animate, animate_index;  // _hscroll, _animate, _animate_index (video.cpp:940-943)
    uint16_t* d_field;
    uint8_t* d_overlay;
    std::vector<uint16_t> field;
    VideoState()
        : ctx(0), ntsc(1), frames(0), front(-1), line_counter(0), frame_counter(0), hscroll(0), animate(0),
          animate_index(0), d_field(0), d_overlay(0)
    {
    }
    void step_animation()  // animate(), video.cpp:1076-1088
    {
        static const int16_t easd[16] = {0, 8, 16, 24, 48, 72, 104, 136, 176, 216, 248, 280, 304, 328, 336, 344};
        if (animate_index == 0)
            hscroll = 0;
        else if (animate_index < 0)
            hscroll = -easd[-++animate_index];
        else
            hscroll = easd[--animate_index];
    }
};
inline VideoState& vs()
{
    static VideoState v;
    return v;
}
}  // namespace efx_player_detail

void video_init(int ntsc)  // video.cpp:572-601
{
    efx_player_detail::VideoState& v = efx_player_detail::vs();
    v.ntsc = ntsc ? 1 : 0;
    efx_video_get_params(v.ntsc, &v.vp);
    if (!v.ctx) {
        efx_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.max_streams = 1;
        cfg.max_pictures = 1;
        cfg.ring_depth = 2;
        cfg.max_stream_bytes = 4096;
        if (efx_create(&cfg, &v.ctx) != EFX_OK) {
            fprintf(stderr, "video_init: efx_create failed (a gfx950 device is required)\n");
            abort();
        }
    }
    if (v.d_field)
        efx_device_free(v.ctx, v.d_field);
    size_t n = (size_t)v.vp.line_width * v.vp.line_count;
    void* p = 0;
    efx_device_alloc(v.ctx, n * 2, &p);
    v.d_field = (uint16_t*)p;
    if (!v.d_overlay) {
        efx_device_alloc(v.ctx, sizeof(_video_composite), &p);
        v.d_overlay = (uint8_t*)p;
    }
    v.field.assign(n, 0);
    v.line_counter = v.frame_counter = 0;
}

void video_reset() {}
void video_pause(int) {}

uint8_t _video_composite[VIDEO_COMPOSITE_HEIGHT * VIDEO_COMPOSITE_WIDTH];
int _video_composite_blend = 0;
int _video_composite_progress = 0;

void efx_video_present(Frame* frames, int front, int mode)
{
    efx_player_detail::VideoState& v = efx_player_detail::vs();
    v.frames = frames;
    v.front = front;
    // the flip the ISR performs in a blanking line (video.cpp:1162-1175)
    if (mode == 2)
        v.animate_index = -16;
    else if (mode == 3)
        v.animate_index = 16;
    if (mode)
        v.step_animation();
}
int efx_video_line_width() { return efx_player_detail::vs().vp.line_width; }
int efx_video_line_count() { return efx_player_detail::vs().vp.line_count; }

extern "C" void video_isr(volatile void* vbuf)  // video.cpp:1122-1198, one call per scan line
{
    efx_player_detail::VideoState& v = efx_player_detail::vs();
    if (!v.ctx)
        return;
    if (v.line_counter == 0 && v.frames && v.front >= 0) {
        // render the whole field on the GPU from the current front Frame (host memory, may have been
        // drawn into by UI code), then hand it out line by line
        std::vector<uint8_t> fr(EFX_FRAME_BYTES);
        for (int k = 0; k < (v.hscroll ? 2 : 1); k++) {  // ring slot 0 = front, 1 = the other Frame of the pair
            const Frame& f = v.frames[k ? v.front ^ 1 : v.front];
            for (int s = 0; s < FB_SLICES; s++)
                memcpy(&fr[(size_t)s * EFX_STRIP_BYTES], f._slices[s], EFX_STRIP_BYTES);
            efx_upload_frame(v.ctx, 0, k, &fr[0]);
        }
        efx_field_opts o;
        memset(&o, 0, sizeof(o));
        o.n_streams = 1;
        o.slot = 0;
        o.other_slot = 1;
        o.ntsc = v.ntsc;
        o.frame_counter = v.frame_counter;
        o.hscroll = v.hscroll;
        if (_video_composite_blend) {
            efx_memcpy_h2d(v.ctx, v.d_overlay, _video_composite, sizeof(_video_composite));
            o.overlay = v.d_overlay;
            o.overlay_blend = _video_composite_blend < -1 ? -1 : _video_composite_blend;
            o.overlay_progress = _video_composite_progress;
        }
        efx_composite_fields_ex(v.ctx, &o, v.d_field);
        efx_memcpy_d2h(v.ctx, &v.field[0], v.d_field, v.field.size() * 2);
    }
    memcpy((void*)vbuf, &v.field[(size_t)v.line_counter * v.vp.line_width], (size_t)v.vp.line_width * 2);
    if (++v.line_counter == v.vp.line_count) {  // video.cpp:1189-1196
        v.line_counter = 0;
        v.frame_counter++;
        if (_video_composite_blend > 0)
            --_video_composite_blend;
        v.step_animation();
    }
}

// ---- audio out ---------------------------------------------------------------------------------

namespace efx_player_detail {
struct AudioState {
    efx_ctx* ctx;
    int16_t* d_pcm;
    int32_t* d_state;
    uint16_t* d_out;
    int beep;
    efx_pdm_sink sink;
    AudioState() : ctx(0), d_pcm(0), d_state(0), d_out(0), beep(0), sink(0) {}
};
inline AudioState& as()
{
    static AudioState a;
    return a;
}
}  // namespace efx_player_detail

void efx_set_pdm_sink(efx_pdm_sink sink) { efx_player_detail::as().sink = sink; }
void beep() { efx_player_detail::as().beep = 5; }

void write_pcm_16(const int16_t* s, int n, int /*channels*/)  // espflix.ino:123-145
{
    efx_player_detail::AudioState& a = efx_player_detail::as();
    if (!a.ctx) {
        efx_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.max_streams = 1;
        cfg.max_pictures = 1;
        cfg.ring_depth = 2;
        cfg.max_stream_bytes = 4096;
        if (efx_create(&cfg, &a.ctx) != EFX_OK) {
            fprintf(stderr, "write_pcm_16: efx_create failed (a gfx950 device is required)\n");
            abort();
        }
        void* p = 0;
        efx_device_alloc(a.ctx, 128 * 2, &p);
        a.d_pcm = (int16_t*)p;
        efx_device_alloc(a.ctx, 12, &p);
        a.d_state = (int32_t*)p;
        efx_device_alloc(a.ctx, 256 * 2, &p);
        a.d_out = (uint16_t*)p;
        int32_t zero[3] = {0, 0, 0};
        efx_memcpy_h2d(a.ctx, a.d_state, zero, sizeof(zero));
    }
    uint16_t words[256];
    int16_t tone[128];
    if (a.beep) {  // five 128-sample bursts of the 32-point sine, >> 2
        for (int i = 0; i < 128; i++)
            tone[i] = (int16_t)(((int16_t)(-32767.0 * __builtin_sin(2 * 3.14159265358979323846 * (i & 31) / 32))) >> 2);
        a.beep--;
        s = tone;
        n = 128;
    }
    if (s) {
        if (n > 128)
            n = 128;
        efx_memcpy_h2d(a.ctx, a.d_pcm, s, (size_t)n * 2);
        efx_pdm(a.ctx, 1, a.d_pcm, n, a.d_state, a.d_out);
        efx_memcpy_d2h(a.ctx, words, a.d_out, (size_t)n * 4);
        for (int i = 2 * n; i < 256; i++)
            words[i] = 0xAAAA;
    } else {
        for (int i = 0; i < 256; i++)
            words[i] = 0xAAAA;  // PDM silence
    }
    if (a.sink)
        a.sink(words, 256);
}

#endif  // EFX_PLAYER_IMPLEMENTATION
#endif  // EFX_PLAYER_HPP
