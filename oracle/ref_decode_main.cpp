// oracle/_ref decode harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Links against the *unmodified* reference translation units player.cpp, streamer.cpp and
// sbc_decoder.cpp (compiled where they lie under /root/reference/src by oracle/Makefile) and
// drives MpegDecoder exactly the way ESPFlix::play_rom does (reference src/espflix.cpp:1043-1058,
// decode_next 723-737): decoder thread in run(), feeder pops empty Buffers, fills 1504 bytes from
// a Streamer ROM source, pushes them full, a zero-length Buffer ends the stream, then waits for
// DECODER_PAUSED.  Every push_video() up-call (reference src/player.cpp:692-702) is captured.
//
// Usage:
//   efx_ref_decode decode <in.ts|@splash|@vmedia> <out.bin|-> [flush]
//        writes every pushed frame as 12*8448 raw bytes (strip order, pad excluded) to out.bin
//        ("-" = no frame dump) and prints one line per frame "F <idx> <pts> <fnv1a64>" plus
//        "CHAIN <n> <hash>" on stderr-independent fd 3 if open, else stderr.
//        "flush" additionally calls flush_picture(1) at the end (as load_poster does) so the
//        last picture is pushed too.
//   efx_ref_decode fixture <@splash|@vmedia> <out.ts>      dump an embedded clip
//   efx_ref_decode bench <nworkers> <list.txt> [repeat]    CPU baseline: decode every TS file
//        named in list.txt with nworkers forked worker PROCESSES (the reference keeps scratch
//        and event state in process globals, src/player.cpp:732, src/streamer.cpp:305-339: one
//        decoder per process), streams dealt round-robin.  Each worker drives ONE MpegDecoder
//        through ONE play: its streams (x repeat) are fed back to back as a single transport
//        stream -- every stream opens with a sequence header and an I picture, so the decoder
//        simply keeps going, as it does across GOPs of a long file -- and a single zero-length
//        Buffer ends it.  That avoids the reference's racy pause/resume hand-shake between
//        plays and measures decoding, not process creation.  Prints
//        "BENCH streams=<n> pictures=<n> seconds=<s> workers=<n> repeat=<n> failed=<n>"
//        (wall time from the first fork to the last exit; inputs pre-loaded in memory).
//   efx_ref_decode tables <out.bin>                        dump zig_zag[64] + scale_dct_q[64]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <time.h>
#include <sys/wait.h>
#include <signal.h>
#include <string>
#include <vector>

#include "player.h"     // reference header (found via -I/root/reference/src)
#include "streamer.h"
#undef printf           // streamer.h redirects printf to the reference's printf_nano

#include "splash.h"     // embedded fixture clips (reference src/splash.h:12, src/vmedia.h:1)
#include "vmedia.h"

extern uint8_t zig_zag[64];        // reference src/player.cpp:150
extern uint8_t scale_dct_q[];      // reference src/player.cpp:161

std::string to_string(int i) { return std::to_string(i); }   // desktop build lacks it (streamer.cpp:150 is ESP-only)

static FILE* g_out = 0;
static FILE* g_log = 0;
static int g_frames = 0;
static uint64_t g_chain = 0xcbf29ce484222325ull;
static int g_quiet = 0;

static uint64_t fnv1a(const uint8_t* p, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

// capture hook: the decode-frame up-call of the reference (src/video.h:49)
void push_video(Frame* f, int front, int64_t pts, int mode)
{
    Frame* fr = &f[front];
    uint64_t h = 0xcbf29ce484222325ull;
    for (int s = 0; s < FB_SLICES; s++) {
        h = fnv1a(fr->_slices[s], FB_STRIDE * FB_SLICE_HEIGHT, h);
        if (g_out) fwrite(fr->_slices[s], 1, FB_STRIDE * FB_SLICE_HEIGHT, g_out);
    }
    uint8_t le[8];
    for (int i = 0; i < 8; i++) le[i] = (uint8_t)(h >> (8 * i));
    g_chain = fnv1a(le, 8, g_chain);
    if (!g_quiet) fprintf(g_log, "F %d %lld %016llx\n", g_frames, (long long)pts, (unsigned long long)h);
    g_frames++;
}
// audio up-call (src/video.h:50): when EFX_REF_AUDIO_OUT names a file, every byte the reference's
// demux hands to push_audio() is appended to it (pins the restatement's efxo_ts_audio_es)
static FILE* g_audio_out = 0;
void push_audio(const uint8_t* data, int len, int64_t, bool)
{
    if (g_audio_out && len > 0)
        fwrite(data, 1, (size_t)len, g_audio_out);
}
void video_reset() {}

static Frame g_fb[2];           // adjacent, as ESPFlix::_frame_buffers (src/espflix.cpp:641)
static MpegDecoder* g_dec = 0;

static void decoder_thread(void*) { g_dec->run(); }

static void decoder_start()
{
    g_fb[0].init();
    g_fb[1].init();
    g_dec = new MpegDecoder(&g_fb[0], &g_fb[1]);
    set_events(DECODER_RUN);                 // before the thread starts: the event word is racy
    start_thread(decoder_thread, 0);
}

// one play_rom() cycle (src/espflix.cpp:1043-1058) on the running decoder
static int play(const uint8_t* data, int len, bool flush)
{
    int before = g_frames;
    Streamer st;
    st.get_rom(data, len);
    g_dec->reset();
    // the desktop event word is a plain int with unlocked read-modify-write and a condition
    // variable that can miss a notify (src/streamer.cpp:305-339): keep signalling until the
    // decoder thread has left pause()
    while (get_events() & DECODER_PAUSED) {
        set_events(DECODER_RUN);
        usleep(20);
    }
    set_events(DECODER_RUN);
    for (;;) {
        Buffer* b = g_dec->pop_empty();
        if (!b) continue;                     // the 0 pause() pushes to unstick a waiting feeder
        int n = (int)st.read(b->data, (int)sizeof(b->data));
        b->len = n;
        g_dec->push_full(b);
        if (!n) break;
    }
    while (!(get_events() & DECODER_PAUSED))  // poll: wait_events() can lose the wake-up
        usleep(20);
    if (flush)
        g_dec->flush_picture(1);
    return g_frames - before;
}

static int decode_rom(const uint8_t* data, int len, bool flush)
{
    decoder_start();
    return play(data, len, flush);
}

static std::vector<uint8_t> slurp(const char* path)
{
    std::vector<uint8_t> v;
    if (!strcmp(path, "@splash")) { v.assign(splash_ts, splash_ts + sizeof(splash_ts)); return v; }
    if (!strcmp(path, "@vmedia")) { v.assign(vmedia, vmedia + sizeof(vmedia)); return v; }
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (n && fread(&v[0], 1, n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

static void silence_stdout()
{
    // the reference chatters through putchar(); keep it off the result channel
    if (!freopen("/dev/null", "w", stdout)) {}
}

static double now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: see header of oracle/ref_decode_main.cpp\n"); return 2; }
    g_log = stderr;
    std::string cmd = argv[1];
    if (cmd == "fixture" && argc == 4) {
        std::vector<uint8_t> v = slurp(argv[2]);
        FILE* f = fopen(argv[3], "wb");
        fwrite(&v[0], 1, v.size(), f);
        fclose(f);
        return 0;
    }
    if (cmd == "tables" && argc == 3) {
        FILE* f = fopen(argv[2], "wb");
        fwrite(zig_zag, 1, 64, f);
        fwrite(scale_dct_q, 1, 64, f);
        fclose(f);
        return 0;
    }
    if (cmd == "decode" && argc >= 4) {
        if (getenv("EFX_REF_AUDIO_OUT"))
            g_audio_out = fopen(getenv("EFX_REF_AUDIO_OUT"), "wb");
        std::vector<uint8_t> v = slurp(argv[2]);
        if (strcmp(argv[3], "-")) g_out = fopen(argv[3], "wb");
        bool flush = argc > 4 && !strcmp(argv[4], "flush");
        silence_stdout();
        int n = decode_rom(&v[0], (int)v.size(), flush);
        fprintf(g_log, "CHAIN %d %016llx\n", n, (unsigned long long)g_chain);
        if (g_out) fclose(g_out);
        if (g_audio_out) fclose(g_audio_out);
        fflush(g_log);
        _exit(0);       // decoder thread is parked in pause(); do not run static destructors under it
    }
    if (cmd == "bench" && argc >= 4) {
        int repeat = argc > 4 ? atoi(argv[4]) : 1;
        if (repeat < 1) repeat = 1;
        int workers = atoi(argv[2]);
        std::vector<std::string> files;
        {
            FILE* f = fopen(argv[3], "r");
            char line[4096];
            while (f && fgets(line, sizeof(line), f)) {
                size_t l = strlen(line);
                while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
                if (l) files.push_back(line);
            }
            if (f) fclose(f);
        }
        std::vector<std::vector<uint8_t> > blobs;
        for (size_t i = 0; i < files.size(); i++) blobs.push_back(slurp(files[i].c_str()));   // pre-load: timing excludes IO
        silence_stdout();
        g_quiet = 1;
        int pfd[2];
        if (pipe(pfd)) return 3;
        if (workers < 1) workers = 1;
        if ((size_t)workers > blobs.size()) workers = (int)blobs.size();
        long pictures = 0;
        double t0 = now();
        for (int w = 0; w < workers; w++) {
            pid_t p = fork();
            if (p == 0) {
                alarm(600);  // watchdog
                g_fb[0].init();
                g_fb[1].init();
                g_dec = new MpegDecoder(&g_fb[0], &g_fb[1]);
                g_dec->reset();
                set_events(DECODER_RUN);
                start_thread(decoder_thread, 0);
                for (int r = 0; r < repeat; r++)
                    for (size_t i = w; i < blobs.size(); i += workers) {
                        const uint8_t* src = &blobs[i][0];
                        size_t left = blobs[i].size();
                        while (left) {
                            Buffer* b = g_dec->pop_empty();
                            size_t n = left < sizeof(b->data) ? left : sizeof(b->data);
                            memcpy(b->data, src, n);
                            b->len = (uint32_t)n;
                            g_dec->push_full(b);
                            src += n;
                            left -= n;
                        }
                    }
                Buffer* b = g_dec->pop_empty();
                b->len = 0;
                g_dec->push_full(b);
                while (!(get_events() & DECODER_PAUSED))
                    usleep(50);
                g_dec->flush_picture(1);
                int32_t res[2] = {g_frames, 0};
                if (write(pfd[1], res, 8) != 8) {}
                _exit(0);
            }
        }
        int crashed = 0;
        for (int w = 0; w < workers; w++) {
            int status;
            wait(&status);
            if (!WIFEXITED(status)) crashed++;
        }
        double t1 = now();
        close(pfd[1]);
        int32_t rr[2];
        long failed = crashed;
        while (read(pfd[0], rr, 8) == 8) { pictures += rr[0]; failed += rr[1]; }
        fprintf(stderr, "BENCH streams=%zu pictures=%ld seconds=%.6f workers=%d repeat=%d failed=%ld\n", blobs.size(), pictures, t1 - t0, workers, repeat, failed);
        return 0;
    }
    fprintf(stderr, "bad command\n");
    return 2;
}
