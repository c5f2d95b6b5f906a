// oracle/_ref SBC harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Links against the *unmodified* reference src/sbc_decoder.cpp and calls sbc_init / sbc_decoder
// (reference src/sbc_decoder.cpp:346-378) the way decode_audio() does (src/video.cpp:962-989).
//
// Usage:
//   efx_ref_sbc decode <frames.bin> <frame_bytes> <out.pcm> [probe]
//        frames.bin = concatenated frames of frame_bytes each; every frame goes through
//        sbc_decoder(&sbc, frame, frame_bytes, pcm, ...).  With `probe` the first frame is
//        decoded once more beforehand, as decode_audio() does to learn the frame size.
//        Output: the int16 PCM of every call (decoded bytes as reported), and on stderr one
//        line "R <index> <return value> <decoded bytes>" per call.
//   efx_ref_sbc tables <out.bin>      SBC_syn_8[128] then SBC_proto_8[80] as int32
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>

#include "sbc_decoder.h"  // reference header

extern const uint32_t SBC_syn_8[128];
extern const uint32_t SBC_proto_8[80];

int main(int argc, char** argv)
{
    if (argc == 3 && !strcmp(argv[1], "tables")) {
        FILE* o = fopen(argv[2], "wb");
        fwrite(SBC_syn_8, 4, 128, o);
        fwrite(SBC_proto_8, 4, 80, o);
        fclose(o);
        return 0;
    }
    if (argc < 5 || strcmp(argv[1], "decode"))
        return 2;
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    std::vector<uint8_t> data;
    uint8_t tmp[4096];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0)
        data.insert(data.end(), tmp, tmp + n);
    fclose(f);
    int fb = atoi(argv[3]);
    bool probe = argc > 5 && !strcmp(argv[5], "probe");
    data.resize(data.size() + 1024, 0);  // the bit reader may look at bytes past a malformed frame
    static SBC_Decode sbc;
    sbc_init(&sbc);
    FILE* o = fopen(argv[4], "wb");
    int16_t pcm[256];
    size_t frames = (data.size() - 1024) / fb;
    int idx = 0;
    for (size_t k = 0; k < frames + (probe ? 1 : 0); k++) {
        size_t fi = probe ? (k ? k - 1 : 0) : k;
        int decoded = 0;
        memset(pcm, 0, sizeof(pcm));
        int r = sbc_decoder(&sbc, &data[fi * fb], fb, pcm, sizeof(pcm), &decoded);
        fprintf(stderr, "R %d %d %d\n", idx++, r, decoded);
        fwrite(pcm, 1, decoded, o);
    }
    fclose(o);
    return 0;
}
