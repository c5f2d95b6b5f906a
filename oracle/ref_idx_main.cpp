// oracle/_ref idx_hdr harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Compiles the *unmodified* reference src/espflix.cpp (textually included from where it lies; the
// class keyword is redefined so that the nested ESPFlix::idx_hdr is reachable) and evaluates the
// player's own index arithmetic, idx_hdr::pts2offset / pts2pts (espflix.cpp:589-627).  Nothing
// else of the player is referenced; the linker discards it (--gc-sections).
//
// Usage: efx_ref_idx <video.idx>   then lines "<pts> <speed>" on stdin -> "<offset> <pts>" lines
#include <algorithm>
#include <condition_variable>
#include <map>
#include <math.h>
#include <mutex>
#include <queue>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
#define class struct
#include "espflix.cpp"  // -I $(REF)/src
#undef class
#undef printf

int main(int argc, char** argv)
{
    if (argc != 2)
        return 2;
    ESPFlix::idx_hdr h;
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(&h, 1, sizeof(h), f) != sizeof(h))
        return 2;
    fclose(f);
    long long pts;
    int speed;
    while (scanf("%lld %d", &pts, &speed) == 2)
        printf("%u %lld\n", h.pts2offset(pts, speed), (long long)h.pts2pts(pts, speed));
    return 0;
}
