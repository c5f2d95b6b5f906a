// Drop-in test: a miniature espflix host written ONLY against the reference's player surface
// (Frame, Buffer, MpegDecoder, push_video, video_init/video_isr, write_pcm_16, event flags) --
// the same calls ESPFlix::play_rom / decode_next / load_poster make (reference
// src/espflix.cpp:723-737,1043-1068) -- compiled against include/efx_player.hpp.
// Prints one line per pushed frame "F <idx> <pts> <fnv>", "U <bytes> <fnv>" for the audio bytes handed
// to push_audio, then "V <fnv>" for one composite field of the last frame, "W" for a slide under the
// overlay, "A <fnv>" for three write_pcm_16 calls and "B <fnv>" for beep() + six more.  The fifth field of an F line
// is the number of Buffers the feeder had pushed when the frame arrived; `paced` as second argument makes the feeder
// slower than the decoder (8 ms per Buffer = 1.5 Mbit/s), the situation of a real-time play.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define EFX_PLAYER_IMPLEMENTATION
#include "efx_player.hpp"

static uint64_t fnv(const uint8_t* p, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

static int g_n = 0;
static volatile int g_fed = 0;  // Buffers the feeder has pushed so far
static Frame* g_frames = 0;
static int g_front = 0;

void push_video(Frame* f, int front, int64_t pts, int mode)
{
    (void)mode;
    uint64_t h = 0xcbf29ce484222325ull;
    for (int s = 0; s < FB_SLICES; s++)
        h = fnv(f[front]._slices[s], FB_STRIDE * FB_SLICE_HEIGHT, h);
    printf("F %d %lld %016llx %d\n", g_n++, (long long)pts, (unsigned long long)h, g_fed);
    g_frames = f;
    g_front = front;
}
static uint64_t g_audio_in = 0xcbf29ce484222325ull;
static long g_audio_bytes = 0;
void push_audio(const uint8_t* d, int n, int64_t, bool)  // the bytes the SBC decoder would receive (video.cpp:1006-1019)
{
    g_audio_in = fnv(d, (size_t)n, g_audio_in);
    g_audio_bytes += n;
}

static uint64_t g_audio = 0xcbf29ce484222325ull;
static void audio_sink(const uint16_t* w, int n) { g_audio = fnv((const uint8_t*)w, (size_t)n * 2, g_audio); }

int main(int argc, char** argv)
{
    if (argc < 2)
        return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    static Frame fb[2];  // adjacent, as ESPFlix::_frame_buffers
    fb[0].init();
    fb[1].init();
    MpegDecoder dec(&fb[0], &fb[1]);
    std::thread t([&] { dec.run(); });
    t.detach();

    dec.reset();
    set_events(DECODER_RUN);
    for (;;) {  // decode_next()
        Buffer* b = dec.pop_empty();
        if (!b)
            continue;
        int n = (int)fread(b->data, 1, sizeof(b->data), f);
        b->len = n;
        g_fed++;
        dec.push_full(b);
        if (!n)
            break;
        if (argc > 2 && !strcmp(argv[2], "paced"))  // a feeder slower than the decoder: a real-time play
            usleep(argc > 3 ? atoi(argv[3]) : 8000);  // (default: 1504 bytes at the service's 1.5 Mbit/s)
    }
    wait_events(DECODER_PAUSED);
    dec.flush_picture(1);  // as load_poster does: show the last picture too
    printf("U %ld %016llx\n", g_audio_bytes, (unsigned long long)g_audio_in);

    video_init(1);
    efx_video_present(g_frames, g_front);
    std::vector<uint16_t> line(efx_video_line_width());
    uint64_t h = 0xcbf29ce484222325ull;
    for (int l = 0; l < efx_video_line_count(); l++) {
        video_isr(&line[0]);
        h = fnv((const uint8_t*)&line[0], line.size() * 2, h);
    }
    printf("V %016llx\n", (unsigned long long)h);

    // slide back to the previous picture (push_video mode 3) under a fading time overlay, as the
    // navigation UI does (espflix.cpp:62-84,873): three more fields
    for (int i = 0; i < VIDEO_COMPOSITE_WIDTH * VIDEO_COMPOSITE_HEIGHT; i++)
        _video_composite[i] = (uint8_t)(i * 7);
    _video_composite_blend = 33;
    _video_composite_progress = 120;
    efx_video_present(g_frames, g_front ^ 1, 3);
    h = 0xcbf29ce484222325ull;
    for (int l = 0; l < 3 * efx_video_line_count(); l++) {
        video_isr(&line[0]);
        h = fnv((const uint8_t*)&line[0], line.size() * 2, h);
    }
    printf("W %016llx %d\n", (unsigned long long)h, _video_composite_blend);

    efx_set_pdm_sink(audio_sink);
    int16_t pcm[128];
    for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 128; i++)
            pcm[i] = (int16_t)((i * 37 + c * 1000) % 4001 - 2000);
        write_pcm_16(c == 1 ? 0 : pcm, 128, 1);
    }
    printf("A %016llx\n", (unsigned long long)g_audio);
    // beep(): the key-press feedback tone, five 128-sample bursts in place of the PCM (espflix.ino:109-135)
    g_audio = 0xcbf29ce484222325ull;
    beep();
    for (int c = 0; c < 6; c++) {
        for (int i = 0; i < 128; i++)
            pcm[i] = (int16_t)((i * 11 + c * 300) % 2001 - 1000);
        write_pcm_16(pcm, 128, 1);
    }
    printf("B %016llx\n", (unsigned long long)g_audio);
    fflush(stdout);
    _exit(0);
}
