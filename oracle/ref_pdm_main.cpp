// oracle/_ref PDM harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The reference's delta-sigma modulator lives in the Arduino sketch espflix.ino, which cannot be
// compiled on a desktop host (ESP-IDF includes at espflix.ino:17-28).  oracle/Makefile therefore
// extracts the text of pdm_second_order() + write_pcm_16() (espflix.ino:73-145) verbatim into
// the git-ignored build directory oracle/_ref/ino_audio_extract.inc at build time and this file
// includes it, so the code that runs is the reference's own, unmodified.  The only shims are
// the ESP-IDF names write_pcm_16 touches (i2s_write, PLOG, portMAX_DELAY).
//
// Usage: efx_ref_pdm <pcm_s16le.bin> <out_u16le.bin> [silence_every N] [beep_at K]
//   feeds the PCM in 128-sample calls to write_pcm_16(s,128,1) and records the 256 u16 words
//   each call hands to i2s_write.  State (_i0,_i1,_i2) persists across calls as in the sketch.
//   "silence_every N": every N-th call passes s==0 (PDM silence 0xAAAA, espflix.ino:139-140).
//   "beep_at K": calls beep() before call K (5 sine bursts, espflix.ino:119-133).
// Usage: efx_ref_pdm bench <pcm_s16le.bin> <repeat> <workers>
//   CPU baseline of bench.py's video_out leg: `workers` processes, each feeds the PCM `repeat` times to
//   write_pcm_16(s,128,1) (output discarded); stderr: BENCH samples=.. seconds=.. workers=..
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#define PLOG(_x)
#define portMAX_DELAY 0
typedef int i2s_port_t;
static FILE* g_o = 0;
static volatile unsigned g_sink = 0;
static int i2s_write(i2s_port_t, const void* src, size_t n, size_t* written, int)
{
    if (g_o)
        fwrite(src, 1, n, g_o);
    else
        g_sink += ((const volatile uint16_t*)src)[n / 2 - 1];  // bench: the words are produced, not kept
    *written = n;
    return 0;
}

#include "ino_audio_extract.inc"

int main(int argc, char** argv)
{
    if (argc == 5 && !strcmp(argv[1], "bench")) {
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 2;
        std::vector<int16_t> pcm;
        int16_t buf[128];
        while (fread(buf, 2, 128, f) == 128) pcm.insert(pcm.end(), buf, buf + 128);
        fclose(f);
        const int repeat = atoi(argv[3]);
        int workers = atoi(argv[4]);
        if (workers < 1) workers = 1;
        const int calls = (int)(pcm.size() / 128);
        struct timeval t0, t1;
        gettimeofday(&t0, 0);
        for (int w = 0; w < workers; w++)
            if (fork() == 0) {
                for (int r = 0; r < repeat; r++)
                    for (int c = 0; c < calls; c++)
                        write_pcm_16(&pcm[c * 128], 128, 1);
                _exit(0);
            }
        for (int w = 0; w < workers; w++) {
            int st;
            wait(&st);
        }
        gettimeofday(&t1, 0);
        fprintf(stderr, "BENCH samples=%ld seconds=%.6f workers=%d\n", (long)calls * 128 * repeat * workers,
                (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec), workers);
        return 0;
    }
    if (argc < 3) return 2;
    int silence_every = 0, beep_at = -1;
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "silence_every")) silence_every = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "beep_at")) beep_at = atoi(argv[i + 1]);
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int16_t> pcm;
    int16_t buf[128];
    while (fread(buf, 2, 128, f) == 128) pcm.insert(pcm.end(), buf, buf + 128);
    fclose(f);
    g_o = fopen(argv[2], "wb");
    int calls = (int)(pcm.size() / 128);
    for (int c = 0; c < calls; c++) {
        if (c == beep_at) beep();
        bool silent = silence_every && (c % silence_every) == silence_every - 1;
        write_pcm_16(silent ? 0 : &pcm[c * 128], 128, 1);
    }
    fclose(g_o);
    return 0;
}
