// efx_tables.cpp -- host-side construction of the look-up tables the kernels use.
//
//  * ParseTables: flat VLC decode tables expanded from the ISO 11172-2 code books in
//    mpeg1_codebook.h (the reference walks bit-serial tree tables instead, player.cpp:516-530,
//    and a prefix-class decoder for DCT coefficients, player.cpp:548-644).
//  * VideoTables: composite-video geometry, sync/burst levels and the chroma-phase LUT, derived
//    with the same float/double arithmetic the reference uses at init time
//    (video_init/pal_init/usec, video.cpp:554-630; LUT derivation gen_palettes,
//    espflix.cpp:1091-1161).  tests/test_abi.py and tests/test_oracle_golden.py pin the values
//    against the reference-derived goldens (geometry, colour LUT, zig-zag, pre-multipliers).
#include <cmath>
#include <cstring>

#include "efx_internal.h"
#include "mpeg1_codebook.h"

namespace efx {

namespace {

template <typename T, int N, typename F>
void expand(T (&table)[N], int index_bits, int code, int len, F make_entry)
{
    // every index whose top `len` bits equal `code` maps to this code word
    int free_bits = index_bits - len;
    int first = code << free_bits;
    for (int i = 0; i < (1 << free_bits); i++)
        table[first + i] = make_entry();
}

}  // namespace

void build_parse_tables(ParseTables* t)
{
    std::memset(t, 0, sizeof(*t));

    // DCT coefficients.  dct_hi covers code words of <= 8 bits (index = their 8-bit prefix);
    // longer words all start with six zero bits and are resolved through dct_lo, indexed by
    // bits 6..15 of a 16-bit peek.
    // (a bit pattern that is no code word: 0 bits consumed, level 63 -- the parser's "this block ends here" mark, which
    // end_of_block carries with 2 bits)
    for (auto& e : t->dct_hi)
        e = (uint16_t)(63 << 10);
    for (auto& e : t->dct_lo)
        e = (uint16_t)(63 << 10);
    for (const DctCode& c : kDctCodes) {
        uint16_t e = (uint16_t)((c.len + 1) | (c.run << 5) | (c.level << 10));  // length incl. the sign bit
        if (c.len <= 8)
            expand(t->dct_hi, 8, c.code, c.len, [&] { return e; });
        else
            expand(t->dct_lo, 10, c.code & ((1 << (c.len - 6)) - 1), c.len - 6, [&] { return e; });
    }
    // level 0 = escape: 6-bit code + 6-bit run + 8-bit level (the 16-bit level form adds 8, in the parser)
    expand(t->dct_hi, 8, kDctEscapeCode, kDctEscapeLen, [&] { return (uint16_t)(kDctEscapeLen + 14); });
    // the two codes starting with 1 (not in the first position of a non-intra block, which the
    // parser handles before its loop): "10" = end_of_block (level 63 marks it), "11s" = (0, +-1)
    expand(t->dct_hi, 8, 0x2, 2, [&] { return (uint16_t)(2 | (63 << 10)); });
    expand(t->dct_hi, 8, 0x3, 2, [&] { return (uint16_t)(3 | (0 << 5) | (1 << 10)); });

    for (const VlcCode& c : kMbaCodes)
        expand(t->mba, 11, c.code, c.len, [&] { return (uint16_t)(c.len | (c.value << 4)); });
    for (const VlcCode& c : kMotionCodes)
        expand(t->motion, 11, c.code, c.len, [&] { return (uint16_t)(c.len | ((c.value + 16) << 4)); });
    for (const VlcCode& c : kCbpCodes)
        expand(t->cbp, 9, c.code, c.len, [&] { return (uint16_t)(c.len | (c.value << 4)); });
    for (const VlcCode& c : kTypePCodes)
        expand(t->type_p, 6, c.code, c.len, [&] { return (uint8_t)(c.len | (c.value << 3)); });

    // IDCT pre-multipliers round(32 s_i s_j), s_0 = 1, s_k = sqrt(2) cos(k pi / 16): the scaled
    // AAN factors the reference tabulates as scale_dct_q (player.cpp:161-170).
    double s[8];
    s[0] = 1.0;
    for (int k = 1; k < 8; k++)
        s[k] = std::sqrt(2.0) * std::cos(k * M_PI / 16);
    for (int n = 0; n < 64; n++) {
        int zz = kZigZag[n];
        int premul = (int)std::floor(32.0 * s[zz >> 3] * s[zz & 7] + 0.5);
        t->scan[n] = (uint32_t)zz | ((uint32_t)premul << 8) | ((uint32_t)kDefaultIntraQ[zz] << 16) | (16u << 24);
    }
}

namespace {

uint32_t ire(double x)  // IRE(), video.cpp:520: DAC code for an IRE level, in the high byte
{
    return ((uint32_t)((x + 40) * 255 / 3.3 / 147.5)) << 8;
}

int usec(float us, float sample_rate, int samples_per_cc)  // video.cpp:554-558
{
    uint32_t r = (uint32_t)(us * sample_rate);
    return (int)(((r + samples_per_cc) / (samples_per_cc << 1)) * (samples_per_cc << 1));
}

int round_half_up(float v)  // RUP(float), espflix.cpp:1071-1077
{
    if (v < 0)
        return -round_half_up(-v);
    return (int)(v + 0.5);
}

// Four carrier-phase samples of one 8-bit chroma value around 2 x black, clipped to 0..127, in
// the blitter's 0,2,1,3 byte order (espflix.cpp:1079-1161).
uint32_t chroma_phases(int c, bool cosine, bool negate)
{
    int black = (int)(ire(7.5) >> 8);
    float scale = (float)black / 33;
    int amp = 128 - c;
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
        double w = cosine ? std::cos(2 * M_PI * i / 4) : std::sin(2 * M_PI * i / 4);
        if (negate)
            w = -w;
        int p = round_half_up(w * amp * scale) + 2 * black;
        p = p < 0 ? 0 : (p < 127 ? p : 127);
        v = (v << 8) | (uint32_t)p;
    }
    return (v & 0xFF0000FFu) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u);
}

}  // namespace

void build_video_tables(int ntsc, VideoTables* t)
{
    std::memset(t, 0, sizeof(*t));
    const int spc = 4;
    t->sync_level = (uint16_t)ire(-40);
    t->blanking_level = (uint16_t)ire(0);
    t->black_level = (uint16_t)ire(7.5);
    if (ntsc) {
        float rate = 315.0 / 88 * spc;
        t->line_width = 228 * spc;
        t->line_count = 262;
        t->hsync_long = usec(63.555 - 4.7, rate, spc);
        t->active_start = usec(10, rate, spc);
        t->hsync = usec(4.7, rate, spc);
        for (int c = 0; c < 256; c++) {
            t->color_tab[c] = chroma_phases(c, false, false);
            t->color_tab[256 + c] = t->color_tab[512 + c] = chroma_phases(c, true, false);
        }
    } else {
        float rate = 4433618.75 * spc / 1000000.0;
        t->pal = 1;
        t->line_width = 284 * spc;
        t->line_count = 312;
        t->hsync_short = usec(2, rate, spc);
        t->hsync_long = usec(30, rate, spc);
        t->hsync = usec(4.7, rate, spc);
        t->burst_start = usec(5.6, rate, spc);
        t->burst_width = (int)(10 * spc + 4) & 0xFFFE;
        t->active_start = usec(10.4, rate, spc);
        uint32_t blank = ire(0);
        float phase = 2 * M_PI / 2;
        for (int i = 0; i < t->burst_width; i++) {
            t->burst0[i] = (int16_t)(blank + std::sin(phase + 3 * M_PI / 4) * blank / 1.5);
            t->burst1[i] = (int16_t)(blank + std::sin(phase - 3 * M_PI / 4) * blank / 1.5);
            phase += 2 * M_PI / spc;
        }
        for (int c = 0; c < 256; c++) {
            t->color_tab[c] = chroma_phases(c, false, false);
            t->color_tab[256 + c] = chroma_phases(c, true, false);
            t->color_tab[512 + c] = chroma_phases(c, true, true);
        }
    }
}

// One sample of a line that carries no picture data at that position (sync 889-893, burst 806-837,
// burst_pal 636-644, blanking 904-914, pal_sync 918-934 of src/video.cpp), as a function of the line
// kind: 0 normal line (sync, burst, black), 1 NTSC vertical blanking, 2 PAL sync lines 304..311.
static uint32_t line_sample(const VideoTables& v, int kind, int i, int line_counter)
{
    if (kind == 0) {
        if (i < v.hsync)
            return v.sync_level;
        if (v.pal) {
            int j = i - v.burst_start;
            if (j >= 0 && j < v.burst_width) {
                const int16_t* b = (line_counter & 1) ? v.burst0 : v.burst1;
                return (uint16_t)b[j ^ 1];
            }
        } else {
            int j = i - v.hsync;
            if (j < 40) {  // 10 cycles of burst, 4 samples per cycle (video.cpp:817-822)
                int ph = j & 3;
                uint32_t bl = v.blanking_level;
                return ph == 0 ? bl + bl / 2 : (ph == 2 ? bl - bl / 2 : bl);
            }
        }
        return v.black_level;
    }
    if (kind == 1)
        return i < v.hsync_long ? v.sync_level : v.blanking_level;
    const uint32_t types = 0x00233000u;  // _sync_type[8] = {0,0,0,3,3,2,0,0}, one nibble each
    int t = (types >> ((line_counter - 1 - 304) * 4)) & 0xF;
    int half = v.line_width / 2;
    int h = i >= half;
    int sw = (t & (h ? 1 : 2)) ? v.hsync_long : v.hsync_short;
    return (i - h * half) < sw ? v.sync_level : v.blanking_level;
}

void build_video_line_templates(const VideoTables* v, VideoLineTemplates* out)
{
    std::memset(out, 0, sizeof(*out));
    for (int i = 0; i < v->line_width; i++) {
        out->tpl[0][i] = (uint16_t)line_sample(*v, 0, i, 1);  // _line_counter odd
        out->tpl[1][i] = (uint16_t)line_sample(*v, 0, i, 2);  // even (differs for PAL only)
        out->tpl[2][i] = (uint16_t)line_sample(*v, 1, i, 0);
        if (v->pal) {
            out->tpl[3][i] = (uint16_t)line_sample(*v, 2, i, 304 + 1);      // type 0
            out->tpl[4][i] = (uint16_t)line_sample(*v, 2, i, 304 + 5 + 1);  // type 2
            out->tpl[5][i] = (uint16_t)line_sample(*v, 2, i, 304 + 3 + 1);  // type 3
        }
    }
}

// SBC synthesis matrix and prototype window (sbc_decoder.cpp:41-71) from their A2DP Appendix B
// definitions: N[i][k] = cos((i + 4)(2k + 1) pi / 16) in 16.16, and the 80-tap prototype filter
// Proto_8_80 scaled by -8 in 1.15, both rounded toward minus infinity.  The filter is symmetric
// about tap 40 except that taps 48 and 64 are the negatives of taps 32 and 16.
void build_sbc_tables(SbcTables* t)
{
    static const double proto_half[41] = {
        0.00000000E+00, 1.56575398E-04, 3.43256425E-04, 5.54620202E-04, 8.23919506E-04, 1.13992507E-03,
        1.47640169E-03, 1.78371725E-03, 2.01182542E-03, 2.10371989E-03, 1.99454554E-03, 1.61656283E-03,
        9.02154502E-04, -1.78805361E-04, -1.64973098E-03, -3.49717454E-03, 5.65949473E-03, 8.02941163E-03,
        1.04584443E-02, 1.27472335E-02, 1.46525263E-02, 1.59045603E-02, 1.62208471E-02, 1.53184106E-02,
        1.29371806E-02, 8.85757540E-03, 2.92408442E-03, -4.91578024E-03, -1.46404076E-02, -2.61098752E-02,
        -3.90751381E-02, -5.31873032E-02, 6.79989431E-02, 8.29847578E-02, 9.75753918E-02, 1.11196689E-01,
        1.23264548E-01, 1.33264415E-01, 1.40753505E-01, 1.45389847E-01, 1.46955068E-01};
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 8; k++) {
            double x = std::cos((i + 4) * (2 * k + 1) * M_PI / 16) * 65536.0;
            t->syn[i * 8 + k] = std::fabs(x) < 1e-6 ? 0 : (int32_t)std::floor(x);
        }
    for (int n = 0; n < 80; n++) {
        int h = n <= 40 ? n : 80 - n;
        double p = proto_half[h];
        if (n > 40 && (h == 16 || h == 32))
            p = -p;
        t->proto[(n & 7) * 10 + (n >> 3)] = p == 0 ? 0 : (int32_t)std::floor(-8.0 * 32768.0 * p);
    }
}

}  // namespace efx
