// tests/dropin_main.cpp -- the PLATFORM side of the drop-in test: everything the reference's host player needs
// that is hardware or pacing (WiFi, IR remote, non-volatile storage, the vsync-paced push_video, the audio
// thread) as benign stubs, plus libefx's player surface (include/espflix_dropin/player.h -> efx_player.hpp).
//
// It is linked with the UNMODIFIED reference sources src/espflix.cpp and src/streamer.cpp (compiled where they
// lie under /root/reference by `make dropin`; src/player.cpp and src/video.cpp are NOT in the build): the
// reference's own ESPFlix::run() -> play_rom(splash_ts) (src/espflix.cpp:1043-1058, decode_next 723-737) drives
// MpegDecoder exactly as on the device, and every push_video() up-call is logged to $EFX_DROPIN_LOG as
// "F <index> <pts> <fnv1a64 of the front Frame>".  The process exits when the play is over.
// (A second build shadows src/splash.h with a generated header so that the same unmodified code plays a
// 1008-picture synthetic stream.)
#include <stdio.h>
#include <unistd.h>

#include <map>
#include <string>

#define EFX_PLAYER_IMPLEMENTATION
#include "player.h"  // include/espflix_dropin/player.h
#undef printf         // (streamer.h redirects printf to the reference's printf_nano)

static FILE* g_log = 0;
static int g_frames = 0;
static long g_audio_bytes = 0;
static volatile long g_nec_polls = 0;

static uint64_t fnv(const uint8_t* p, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

// ---- the up-calls of src/video.cpp, without its vsync pacing (out of scope) ---------------
void push_video(Frame* f, int front, int64_t pts, int mode)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int s = 0; s < FB_SLICES; s++)
        h = fnv(f[front]._slices[s], FB_STRIDE * FB_SLICE_HEIGHT, h);
    fprintf(g_log, "F %d %lld %016llx\n", g_frames++, (long long)pts, (unsigned long long)h);
    efx_video_present(f, front, mode);
}
void push_audio(const uint8_t*, int len, int64_t, bool) { g_audio_bytes += len; }
extern "C" void audio_thread(void*)
{
    for (;;)
        usleep(100000);
}

// ---- hardware the player polls -----------------------------------------------------------------------------------
std::map<std::string, int>& wifi_list()
{
    static std::map<std::string, int> none;
    return none;
}
void wifi_join(const char*, const char*) {}
std::string wifi_ssid() { return ""; }
WiFiState wifi_state() { return NONE; }
void wifi_scan() {}
void wifi_disconnect() {}
void nv_write(const char*, int64_t) {}
int64_t nv_read(const char*) { return 0; }
void up_key() {}
void down_key() {}
std::string to_string(int i) { return std::to_string(i); }

// The IR poll (src/espflix.cpp:1020-1040) is the player's idle loop: once the decoder has parked at the end of
// the play, the run is over.
int get_nec()
{
    g_nec_polls++;
    if (g_frames > 0 && (get_events() & DECODER_PAUSED) && !(get_events() & DECODER_RUN)) {
        fprintf(g_log, "DONE %d %ld\n", g_frames, g_audio_bytes);
        fflush(g_log);
        _exit(0);
    }
    return 0;
}

// watchdog: a play that stalls is reported (with what the event word looked like), not waited for
static void watchdog()
{
    const char* t = getenv("EFX_DROPIN_TIMEOUT");
    const int limit = t ? atoi(t) : 120;
    int last = -1, idle = 0;
    for (;;) {
        sleep(1);
        if (g_frames != last) {
            last = g_frames;
            idle = 0;
        } else if (++idle >= limit) {
            fprintf(g_log, "HANG frames=%d events=%x nec_polls=%ld audio=%ld\n", g_frames, get_events(), g_nec_polls, g_audio_bytes);
            fflush(g_log);
            _exit(3);
        }
    }
}

int main()
{
    const char* path = getenv("EFX_DROPIN_LOG");
    g_log = path ? fopen(path, "w") : stderr;
    if (!g_log)
        return 2;
    new std::thread(watchdog);
    if (!freopen("/dev/null", "w", stdout)) {}  // the reference chatters through printf_nano
    espflix_run(1);                             // src/espflix.cpp:1210: video_init, new ESPFlix, run() -- never returns
    return 1;
}
