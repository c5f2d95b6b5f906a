// efx_tables.cpp -- host-side construction of the look-up tables the kernels use.
//
//  * ParseTables: the zig-zag / pre-multiplier / default-matrix table of k_index and k_recon.
//  * TmTables (build_tm_tables, below): the slice parser's token-machine tables, expanded from the ISO 11172-2 code books
//    in mpeg1_codebook.h (the reference walks bit-serial tree tables instead, player.cpp:516-530, and a prefix-class
//    decoder for DCT coefficients, player.cpp:548-644).
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

void build_parse_tables(ParseTables* t)
{
    std::memset(t, 0, sizeof(*t));

    // IDCT pre-multipliers round(32 s_i s_j), s_0 = 1, s_k = sqrt(2) cos(k pi / 16): the scaled
    // AAN factors the reference tabulates as scale_dct_q (player.cpp:161-170).
    double s[8];
    s[0] = 1.0;
    for (int k = 1; k < 8; k++)
        s[k] = std::sqrt(2.0) * std::cos(k * M_PI / 16);
    (void)s;  // (k_recon scales by the pre-multipliers as compile-time constants: premul_at(), k_recon.hip)
    for (int n = 0; n < 64; n++) {
        int zz = kZigZag[n];
        // Where raster position zz = 8 r + c sits in k_recon's per-lane block of 64 int16 (round 6): column by column, and
        // inside a column the rows in the PAIRS the butterfly's first stage combines -- (0,4) (2,6) (1,7) (3,5), one pair per
        // dword -- so that a dword read + one v_dot2_i32_i16 with the two pre-multipliers packed in a constant delivers
        // x_a p_a + x_b p_b (or the difference) directly.
        static const int kRowSlot[8] = {0, 4, 2, 6, 1, 7, 3, 5};  // row -> 2 * pair + half
        int slot = (zz & 7) * 8 + kRowSlot[zz >> 3];
        t->scan[n] = (uint32_t)zz | ((uint32_t)slot << 8) | ((uint32_t)kDefaultIntraQ[zz] << 16) | (16u << 24);
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
    for (int bits = 0; bits < 20; bits++) {
        const uint64_t d = bits >= 1 && bits <= 16 ? (1ull << bits) - 1 : 1;
        t->iq_magic[bits] = d == 1 ? 1u : (uint32_t)((1ull << 32) / d + 1);  // (2^l - d = 1 for d = 2^l - 1)
    }
}

}  // namespace efx

// ---- the token machine's tables (parse_tm.h) -----------------------------------------------------------------------------
#include "parse_tm.h"

namespace efx {

namespace {

struct TmBuilder {
    TmTables* t;
    // entry: code bits, raw bits, value, run / header value, flags, hacc shift of this token, next state word
    static TmE make(int len, int xb, int v10, int r6, uint32_t flags, int hsh, uint32_t next)
    {
        TmE e;
        e.x = (uint32_t)len | ((uint32_t)(v10 & 0x3FF) << 6) | ((uint32_t)xb << 16) | ((uint32_t)r6 << 20) | flags;
        e.y = next | ((uint32_t)hsh << 6);
        return e;
    }
    // every index of the `peek`-bit table at `base` whose top `len` bits equal `code`
    void fill(int base, int peek, uint32_t code, int len, TmE e)
    {
        const int free_bits = peek - len;
        for (int i = 0; i < (1 << free_bits); i++)
            t->e[base + (int)(code << free_bits) + i] = e;
    }
};

}  // namespace

void build_tm_tables(TmTables* t)
{
    TmBuilder B{t};
    // every table starts out as "no such code": nothing consumed, nothing changed, on to the dead state of its kind
    auto dead_fill = [&](int base, int count, uint32_t why) {
        for (int i = 0; i < count; i++)
            t->e[base + i] = TmBuilder::make(0, 0, 0, 0, 0, kHaccNone, tm_dead(why));
    };
    dead_fill(0, kTmEntries, kDeadBadBlock);
    dead_fill(kTbMvH1, 256, kDeadBadHeader);
    dead_fill(kTbCbp1, 256, kDeadBadHeader);
    dead_fill(kTbTypeP, 64, kDeadBadHeader);
    dead_fill(kTbTypeI, 64, kDeadBadHeader);
    dead_fill(kTbMvH2, 64, kDeadBadHeader);
    dead_fill(kTbMvV2, 64, kDeadBadHeader);
    dead_fill(kTbMvV1, 256, kDeadBadHeader);
    dead_fill(kTbCbp2, 6, kDeadBadHeader);
    dead_fill(kTbMbaA1, 256, kDeadBadMba);
    dead_fill(kTbMbaB1, 256, kDeadBadMba);
    dead_fill(kTbMbaA2, 64, kDeadBadMba);
    dead_fill(kTbMbaB2, 64, kDeadBadMba);
    dead_fill(kTbEnd3, 48, kDeadBadMba);
    for (uint32_t why = 0; why < 8; why++)
        dead_fill(kTbDead + 2 * (int)why, 2, why);  // a dead state leads to itself
    const uint32_t wTypeP = tm_word(kTbTypeP, 6), wMbaA2 = tm_word(kTbMbaA2, 6), wMbaB1 = tm_word(kTbMbaB1, 8),
                   wMbaB2 = tm_word(kTbMbaB2, 6), wMvH2 = tm_word(kTbMvH2, 6), wMvV1 = tm_word(kTbMvV1, 8),
                   wMvV2 = tm_word(kTbMvV2, 6), wDctLo = tm_word(kTbDctLo, 10), wEscR = tm_word(kTbEscR, 6),
                   wEscL = tm_word(kTbEscL, 8), wEscP = tm_word(kTbEscP, 1), wEscN = tm_word(kTbEscN, 1),
                   wDcY2 = tm_word(kTbDcY2, 1), wDcC2 = tm_word(kTbDcC2, 2);
    const int none = kHaccNone;

    // macroblock_address_increment (table B-1) with stuffing (34) and escape (35), player.cpp:1267-1275.  Code words of
    // 10 and 11 bits start with five zero bits: the 8-bit first level hands them to a 6-bit second level.  The B tables
    // are the reference's SECOND loop, which only looks for escapes: after an escape a stuffing code counts as 34.
    for (int after_escape = 0; after_escape < 2; after_escape++) {
        const int l1 = after_escape ? kTbMbaB1 : kTbMbaA1, l2 = after_escape ? kTbMbaB2 : kTbMbaA2;
        for (const VlcCode& c : kMbaCodes) {
            TmE e;
            if (c.value == 35)
                e = TmBuilder::make(0, 0, 0, 33, 0, kHaccInc, wMbaB1);
            else if (c.value == 34 && !after_escape)
                e = TmBuilder::make(0, 0, 0, 1, 0, kHaccStuff, kWMbaA1);
            else
                e = TmBuilder::make(0, 0, 0, c.value, kTmTypeFix, kHaccInc, wTypeP);
            if (c.len <= 8) {
                e.x |= (uint32_t)c.len;
                B.fill(l1, 8, c.code, c.len, e);
            } else {
                e.x |= (uint32_t)(c.len - 5);
                B.fill(l2, 6, c.code & ((1u << (c.len - 5)) - 1), c.len - 5, e);
            }
        }
        for (int i = 0; i < 6; i++)  // 0000 0xxx, xxx < 110: five zero bits, the rest at the second level
            t->e[l1 + i] = TmBuilder::make(5, 0, 0, 0, 0, none, after_escape ? wMbaB2 : wMbaA2);
    }
    // slice_done(), player.cpp:1238-1249: 23 zero bits where an address increment would start end the slice.  11 of them
    // have brought the lane to entry 0 of the second level; 4 + 4 + 4 more lead to the dead state kDeadEnd (tm_finish
    // checks that no stuffing came before them: the reference looks for the end of the slice only between macroblocks).
    t->e[kTbMbaA2] = TmBuilder::make(6, 0, 0, 0, 0, none, tm_word(kTbEnd3, 4));
    t->e[kTbEnd3] = TmBuilder::make(4, 0, 0, 0, 0, none, tm_word(kTbEnd4, 4));
    t->e[kTbEnd4] = TmBuilder::make(4, 0, 0, 0, 0, none, tm_word(kTbEnd5, 4));
    t->e[kTbEnd5] = TmBuilder::make(0, 0, 0, 0, 0, none, tm_dead(kDeadEnd));

    // macroblock_type, tables B-2a / B-2b (player.cpp:1292-1296): bit0 intra, bit1 pattern, bit3 motion forward, bit4 quant
    // (+ 5 bits of quantiser_scale).  What follows is the macroblock's plan: motion, pattern, blocks 0..5 from bit 9 down.
    auto type_entry = [&](int len, int value) {
        const int plan = (value & 1) ? 0x3F : (((value & 8) ? 0x80 : 0) | ((value & 2) ? 0x40 : 0));
        return TmBuilder::make(len, (value & 0x10) ? 5 : 0, plan << 2, value, kTmQuery | kTmType, 0, 0);
    };
    for (const VlcCode& c : kTypePCodes)
        B.fill(kTbTypeP, 6, c.code, c.len, type_entry(c.len, c.value));
    for (const VlcCode& c : kTypeICodes)
        B.fill(kTbTypeI, 6, c.code, c.len, type_entry(c.len, c.value));

    // motion codes, table B-4 (motion_vector(), player.cpp:891-910): the code + 16 and the forward_r_size residual bits
    // that follow every code but 0 go into the macroblock's header word.  Horizontal, then vertical; after the vertical one
    // the plan decides (pattern, or the macroblock is complete).
    for (int vertical = 0; vertical < 2; vertical++) {
        const int l1 = vertical ? kTbMvV1 : kTbMvH1, l2 = vertical ? kTbMvV2 : kTbMvH2;
        for (const VlcCode& c : kMotionCodes) {
            TmE e = TmBuilder::make(0, 0, 0, c.value + 16, (c.value ? kTmR : 0) | (vertical ? kTmQuery : 0),
                                    vertical ? kHaccMvV : kHaccMvH, vertical ? 0u : wMvV1);
            if (c.len <= 8) {
                e.x |= (uint32_t)c.len;
                B.fill(l1, 8, c.code, c.len, e);
            } else {
                e.x |= (uint32_t)(c.len - 5);
                B.fill(l2, 6, c.code & ((1u << (c.len - 5)) - 1), c.len - 5, e);
            }
        }
        for (int i = 0; i < 6; i++)
            t->e[l1 + i] = TmBuilder::make(5, 0, 0, 0, 0, none, vertical ? wMvV2 : wMvH2);
    }

    // coded_block_pattern, table B-3 (player.cpp:1307): the blocks join the plan.  The six 9-bit codes share three 8-bit
    // prefixes; each prefix has its own two-entry second level.
    for (const VlcCode& c : kCbpCodes) {
        if (c.len <= 8)
            B.fill(kTbCbp1, 8, c.code, c.len, TmBuilder::make(c.len, 0, c.value << 4, 0, kTmQuery, none, 0));
        else {
            const int prefix = c.code >> 1;  // 1, 2 or 3
            t->e[kTbCbp1 + prefix] = TmBuilder::make(8, 0, 0, 0, 0, none, tm_word(kTbCbp2 + 2 * (prefix - 1), 1));
            t->e[kTbCbp2 + 2 * (prefix - 1) + (c.code & 1)] = TmBuilder::make(1, 0, c.value << 4, 0, kTmQuery, none, 0);
        }
    }

    // dct_dc_size (player.cpp:1010-1068 as it reads them, beyond tables B-5a / B-5b): the size is the value AND the
    // number of differential bits that follow.  Luminance: 00 -> 1, 01 -> 2, 100 -> 0, 101 -> 3, then k >= 2 ones and
    // a zero -> k + 2 in k + 1 bits.  Chrominance: 00 -> 0, 01 -> 1, k ones and a zero -> k + 1 in min(k + 1, 10) bits.
    auto dc_entry = [&](int len, int size) { return TmBuilder::make(len, size, size, 0, kTmEmit, none, kWDct); };
    B.fill(kTbDcY1, 8, 0x0, 2, dc_entry(2, 1));
    B.fill(kTbDcY1, 8, 0x1, 2, dc_entry(2, 2));
    B.fill(kTbDcY1, 8, 0x4, 3, dc_entry(3, 0));
    B.fill(kTbDcY1, 8, 0x5, 3, dc_entry(3, 3));
    for (int k = 2; k <= 7; k++)
        B.fill(kTbDcY1, 8, (1u << (k + 1)) - 2, k + 1, dc_entry(k + 1, k + 2));
    t->e[kTbDcY1 + 0xFF] = TmBuilder::make(8, 0, 0, 0, 0, none, wDcY2);
    t->e[kTbDcY2 + 0] = dc_entry(1, 10);  // eight ones and a zero
    t->e[kTbDcY2 + 1] = dc_entry(2, 11);  // nine ones: ten bits are consumed whatever the tenth is
    B.fill(kTbDcC1, 8, 0x0, 2, dc_entry(2, 0));
    B.fill(kTbDcC1, 8, 0x1, 2, dc_entry(2, 1));
    for (int k = 1; k <= 7; k++)
        B.fill(kTbDcC1, 8, (1u << (k + 1)) - 2, k + 1, dc_entry(k + 1, k + 1));
    t->e[kTbDcC1 + 0xFF] = TmBuilder::make(8, 0, 0, 0, 0, none, wDcC2);
    t->e[kTbDcC2 + 0] = t->e[kTbDcC2 + 1] = dc_entry(1, 9);  // eight ones and a zero
    t->e[kTbDcC2 + 2] = dc_entry(2, 10);                      // nine ones and a zero
    t->e[kTbDcC2 + 3] = dc_entry(2, 11);                      // ten ones: ten bits consumed

    // DCT coefficients, tables B-5c..f (player.cpp:1070-1103, 532-644): run / level with the sign as one raw bit.  Code
    // words of more than 8 bits start with six zero bits: the first level consumes those, a 10-bit second level has the
    // rest.  "10" = end_of_block: the plan decides; as the FIRST coefficient of a non-intra block "1s" is (0, +-1).  The
    // escape 000001 is followed by a 6-bit run and a level of one byte, or 0x00 / 0x80 and a second byte.
    const uint32_t coef = kTmEmit | kTmCoef;
    for (int first = 0; first < 2; first++) {
        const int hi = first ? kTbDctF : kTbDct;
        for (const DctCode& c : kDctCodes) {
            if (c.len <= 8)
                B.fill(hi, 8, c.code, c.len, TmBuilder::make(c.len, 1, c.level, c.run, coef, none, kWDct));
            else if (!first)
                B.fill(kTbDctLo, 10, c.code & ((1u << (c.len - 6)) - 1), c.len - 6,
                       TmBuilder::make(c.len - 6, 1, c.level, c.run, coef, none, kWDct));
        }
        B.fill(hi, 8, 0x0, 6, TmBuilder::make(6, 0, 0, 0, 0, none, wDctLo));
        B.fill(hi, 8, kDctEscapeCode, kDctEscapeLen, TmBuilder::make(kDctEscapeLen, 0, 0, 0, 0, none, wEscR));
        if (first)
            B.fill(hi, 8, 0x1, 1, TmBuilder::make(1, 1, 1, 0, coef, none, kWDct));
        else {
            B.fill(hi, 8, 0x2, 2, TmBuilder::make(2, 0, 0, 0, kTmQuery, none, 0));
            B.fill(hi, 8, 0x3, 2, TmBuilder::make(2, 1, 1, 0, coef, none, kWDct));
        }
        // A short code word, its sign AND the end_of_block behind it, where all of that fits the 8-bit peek, are ONE token
        // (the signed level in the entry, no raw bit): most blocks end on a short code, and an end_of_block of its own
        // was one trip in seven.
        auto with_eob = [&](uint32_t code, int len, int run, int level) {
            if (len + 3 > 8)
                return;
            for (uint32_t sign = 0; sign < 2; sign++)
                B.fill(hi, 8, (((code << 1) | sign) << 2) | 0x2, len + 3,
                       TmBuilder::make(len + 3, 0, sign ? -level : level, run, coef | kTmQuery, none, kWDct));
        };
        for (const DctCode& c : kDctCodes)
            with_eob(c.code, c.len, c.run, c.level);
        if (first)
            with_eob(0x1, 1, 0, 1);
        else
            with_eob(0x3, 2, 0, 1);
    }
    for (int run = 0; run < 64; run++)
        t->e[kTbEscR + run] = TmBuilder::make(6, 0, 0, run, 0, none, wEscL);
    for (int b = 0; b < 256; b++) {
        if (b == 0x00)
            t->e[kTbEscL + b] = TmBuilder::make(8, 0, 0, 0, 0, none, wEscP);
        else if (b == 0x80)
            t->e[kTbEscL + b] = TmBuilder::make(8, 0, 0, 0, 0, none, wEscN);
        else
            t->e[kTbEscL + b] = TmBuilder::make(8, 0, (int8_t)b, 0, coef, none, kWDct);
    }
    // second byte of a 16-bit level: its top bit chooses the entry, the other seven are raw bits ADDED to the value
    t->e[kTbEscP + 0] = TmBuilder::make(1, 7, 0, 0, coef | kTmAdd, none, kWDct);
    t->e[kTbEscP + 1] = TmBuilder::make(1, 7, 128, 0, coef | kTmAdd, none, kWDct);
    t->e[kTbEscN + 0] = TmBuilder::make(1, 7, -256, 0, coef | kTmAdd, none, kWDct);
    t->e[kTbEscN + 1] = TmBuilder::make(1, 7, -128, 0, coef | kTmAdd, none, kWDct);
}

}  // namespace efx
