// oracle/_ref video-out harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Links against the *unmodified* reference src/video.cpp (force-included oracle/ref_shim.h),
// player.cpp (for Frame), streamer.cpp and sbc_decoder.cpp, and drives the per-scan-line
// callback video_isr() (reference src/video.cpp:1122-1198) the way the I2S DMA interrupt does
// (src/video.cpp:51-56,172-188): once per line, over TWO alternating zero-initialised line
// buffers.  The displayed frame is published through the reference's own file-scope globals
// _frames/_current_frame (src/video.cpp:938,943).
//
// Usage:
//   efx_ref_video field <frames.bin> <ntsc:1|0> <nfields> <out.bin>
//        frames.bin = 2 x 101376 bytes (Frame[0], Frame[1] in strip order); front = 0.
//        Output: nfields x line_count x line_width little-endian u16 samples.
//        _frame_counter starts at 0 and is advanced by the ISR itself (dither parity).
//   efx_ref_video fieldx <frames.bin> <ntsc:1|0> <nfields> <out.bin> <front:0|1> <hscroll.bin|-> <overlay.bin|-> <blend> <progress>
//        the same with the reference's own display state set before the fields run:
//        _current_frame = front; _hscroll (src/video.cpp:941) = the fld-th int16 of hscroll.bin
//        before each field (the ISR's animate() zeroes it again at field end while
//        _animate_index is 0); _video_composite (src/video.cpp:841) = the 1280 overlay bytes,
//        _video_composite_blend / _video_composite_progress as given (the ISR decrements the
//        blend once per field itself).
//   efx_ref_video bench <frames.bin> <ntsc:1|0> <nfields> <workers>
//        CPU baseline of bench.py's video_out leg: `workers` processes, each runs video_isr() for nfields fields
//        of the two frames (lines kept in memory, nothing written); stderr: BENCH fields=.. seconds=.. workers=..
//   efx_ref_video params <ntsc:1|0> <out.bin>
//        dumps int32 {line_width,line_count,hsync,hsync_long,hsync_short,burst_start,
//        burst_width,active_start} followed by _color_tab[768] (u32) and dither4x4[8] (u32).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include "video.h"      // reference header
#include "streamer.h"
#undef printf

extern "C" void video_isr(volatile void* buf);
extern Frame* _frames;
extern int8_t _current_frame;
extern volatile int _line_counter, _frame_counter;
extern int _line_width, _line_count, _hsync, _hsync_long, _hsync_short, _burst_start, _burst_width, _active_start;
extern uint32_t _color_tab[256 * 3];
extern uint32_t dither4x4[];
extern int16_t _hscroll, _animate_index;
extern int _video_composite_blend, _video_composite_progress;

std::string to_string(int i) { return std::to_string(i); }
void video_init_hw(int, int) {}
void ir_sample() {}
void write_pcm_16(const int16_t*, int, int) {}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::string cmd = argv[1];
    if (!freopen("/dev/null", "w", stdout)) {}
    if (cmd == "params" && argc == 4) {
        video_init(atoi(argv[2]));
        int32_t p[8] = { _line_width, _line_count, _hsync, _hsync_long, _hsync_short, _burst_start, _burst_width, _active_start };
        FILE* f = fopen(argv[3], "wb");
        fwrite(p, 4, 8, f);
        fwrite(_color_tab, 4, 768, f);
        fwrite(dither4x4, 4, 8, f);
        fclose(f);
        return 0;
    }
    if (cmd == "bench" && argc == 6) {
        static Frame fb[2];
        fb[0].init();
        fb[1].init();
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 2;
        for (int k = 0; k < 2; k++)
            for (int s = 0; s < FB_SLICES; s++)
                if (fread(fb[k]._slices[s], 1, FB_STRIDE * FB_SLICE_HEIGHT, f) != FB_STRIDE * FB_SLICE_HEIGHT) return 2;
        fclose(f);
        video_init(atoi(argv[3]));
        const int nfields = atoi(argv[4]);
        int workers = atoi(argv[5]);
        if (workers < 1) workers = 1;
        _frames = fb;
        _current_frame = 0;
        struct timeval t0, t1;
        gettimeofday(&t0, 0);
        for (int w = 0; w < workers; w++)
            if (fork() == 0) {
                std::vector<uint16_t> a(_line_width, 0), b(_line_width, 0);
                for (int fld = 0; fld < nfields; fld++)
                    for (int l = 0; l < _line_count; l++)
                        video_isr((l & 1) ? &b[0] : &a[0]);
                _exit(0);
            }
        for (int w = 0; w < workers; w++) {
            int st;
            wait(&st);
        }
        gettimeofday(&t1, 0);
        fprintf(stderr, "BENCH fields=%ld seconds=%.6f workers=%d\n", (long)nfields * workers,
                (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec), workers);
        return 0;
    }
    if ((cmd == "field" && argc == 6) || (cmd == "fieldx" && argc == 11)) {
        const bool ex = cmd == "fieldx";
        static Frame fb[2];
        fb[0].init();
        fb[1].init();
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 2;
        for (int k = 0; k < 2; k++)
            for (int s = 0; s < FB_SLICES; s++)
                if (fread(fb[k]._slices[s], 1, FB_STRIDE * FB_SLICE_HEIGHT, f) != FB_STRIDE * FB_SLICE_HEIGHT) return 2;
        fclose(f);
        video_init(atoi(argv[3]));
        int nfields = atoi(argv[4]);
        _frames = fb;
        _current_frame = 0;
        std::vector<int16_t> hs;
        if (ex) {
            _current_frame = (int8_t)atoi(argv[6]);
            if (strcmp(argv[7], "-")) {
                hs.resize(nfields, 0);
                FILE* h = fopen(argv[7], "rb");
                if (!h || fread(&hs[0], 2, nfields, h) != (size_t)nfields) return 2;
                fclose(h);
            }
            if (strcmp(argv[8], "-")) {
                FILE* h = fopen(argv[8], "rb");
                if (!h || fread(_video_composite, 1, VIDEO_COMPOSITE_WIDTH * VIDEO_COMPOSITE_HEIGHT, h) !=
                              VIDEO_COMPOSITE_WIDTH * VIDEO_COMPOSITE_HEIGHT) return 2;
                fclose(h);
            }
            _video_composite_blend = atoi(argv[9]);
            _video_composite_progress = atoi(argv[10]);
        }
        std::vector<uint16_t> a(_line_width, 0), b(_line_width, 0);
        FILE* o = fopen(argv[5], "wb");
        for (int fld = 0; fld < nfields; fld++) {
            if (!hs.empty()) {
                _animate_index = 0;
                _hscroll = hs[fld];
            }
            for (int l = 0; l < _line_count; l++) {
                uint16_t* buf = (l & 1) ? &b[0] : &a[0];   // two DMA descriptors ping-pong (src/video.cpp:171-186)
                video_isr(buf);
                fwrite(buf, 2, _line_width, o);
            }
        }
        fclose(o);
        return 0;
    }
    return 2;
}
