// efx_gen.cpp -- deterministic synthetic MPEG-1 (ISO 11172-2) I/P stream generator.
//
// Produces the workloads of SURVEY.md section 8(d): 352x192 elementary streams made of
// sequence + GOP headers and I / P pictures, one slice per macroblock row by default, that
// honour every constraint the reference decoder needs (SURVEY section 8a "stream constraints",
// reference src/player.cpp:646-730,1238-1316): 352x192 only, I and P pictures only, the first
// macroblock of each slice is coded, motion vectors keep every fetch inside the picture, and a
// PES PTS precedes each picture when the stream is TS-wrapped (PID 0x100).
//
// It is a small real encoder: forward DCT, quantisation, VLC emission, and a local decoder loop
// (dequantise + integer IDCT + clamp exactly as the reference reconstructs, player.cpp:922-1236)
// so that P pictures are predicted from what a decoder will really hold.  No ffmpeg exists in
// this environment.  This is workload tooling for bench.py and tests, not part of the decode
// path.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../csrc/mpeg1_codebook.h"

namespace {

using efx::kDctCodes;
using efx::kZigZag;

constexpr int W = 352, H = 192, MBW = 22, MBH = 12, CW = 176, CH = 96;

enum : uint32_t {
    FLAG_I_ONLY = 1,         // every picture is an I picture
    FLAG_CUSTOM_MATRICES = 2,  // sequence header loads both quantiser matrices
    FLAG_WIDE_SLICES = 4,    // slices span 2-3 macroblock rows (5 slices, like ffmpeg's output)
    FLAG_LONG_SKIPS = 8,     // P pictures contain skip runs > 33 (macroblock_escape) and stuffing
    FLAG_FLAT_BRIGHT = 16,   // source has flat 255 areas (unclamped DC-only intra blocks)
    FLAG_RATE_1500K = 32,    // quantiser chosen per stream so that the mean picture is ~6.25 kB: the service's
                             // 1.5 Mbit/s at 30 Hz (reference indexer/indexer.cpp:306-309)
    FLAG_HUGE_LEVELS = 64,   // quantiser_scale 1 (with jumps to 31) on a full-range step / checker / noise source: AC levels
                             // up to +-255 and -256, i.e. the 16-bit escape forms "000001 run 00 xx" and "... 80 xx"
                             // (player.cpp:1092-1099); escapes are also forced on levels that have a code, in both
                             // forms, incl. a coded level 0 and runs beyond 31
    FLAG_ODD_HEADERS = 128,  // header quirks (player.cpp:704-724,1328-1330): some P-coded pictures carry a B / D / reserved
                             // picture_coding_type (header ignored, slices decoded with the P books and the f_code of the
                             // last real P header, which varies from picture to picture); user_data and extension start
                             // codes after sequence / GOP / picture headers and between slices.  Everything the
                             // reference's bit-serial marker hunt (player.cpp:1360-1363) walks through is shaped so that
                             // the hunt arrives at the next real start code: zero bits only after an ignored header,
                             // user data in 4-byte groups whose last byte is a harmless marker value
    FLAG_SLICE_EXTRA = 256,  // slice headers carry 0-3 extra_bit_slice = 1 + extra_information_slice bytes before the
                             // closing 0 bit (player.cpp:1261-1262: `while (get_bit()) get_bits(8);`), on I and P pictures;
                             // the bytes have their top bit set, so they cannot take part in a start code
};
constexpr int kCodedZero = 1 << 20;  // levels[]: a coefficient coded with level 0 (escape "00 00"): decoded, not skipped

struct Lcg {
    uint32_t s;
    uint32_t next() { return s = s * 1664525u + 1013904223u; }
};

// ---------------------------------------------------------------------------------------
// bit writer

struct BitWriter {
    std::vector<uint8_t> buf;
    uint64_t acc = 0;
    int n = 0;
    int zrun = 0;  // zero bits at the end of what has been written
    void put(uint32_t v, int len)
    {
        v &= (len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1);
        if (v == 0)
            zrun += len;
        else
            zrun = __builtin_ctz(v);
        acc = (acc << len) | v;
        n += len;
        while (n >= 8) {
            buf.push_back((uint8_t)(acc >> (n - 8)));
            n -= 8;
        }
    }
    void align()
    {
        if (n)
            put(0, 8 - n);
    }
    void start_code(int code)
    {
        align();
        buf.push_back(0);
        buf.push_back(0);
        buf.push_back(1);
        buf.push_back((uint8_t)code);
    }
};

// ---------------------------------------------------------------------------------------
// encode-side views of the code books

struct Books {
    uint16_t mba_code[36];
    uint8_t mba_len[36];
    uint16_t cbp_code[64];
    uint8_t cbp_len[64];
    uint16_t mv_code[33];  // index = code + 16
    uint8_t mv_len[33];
    uint16_t type_p_code[32];
    uint8_t type_p_len[32];
    uint16_t dct_code[32][41];
    uint8_t dct_len[32][41];  // 0 = needs escape
    float cosv[8][8];
    uint8_t premul[64];
    Books()
    {
        memset(this, 0, sizeof(*this));
        for (auto& c : efx::kMbaCodes) {
            mba_code[c.value] = c.code;
            mba_len[c.value] = c.len;
        }
        for (auto& c : efx::kCbpCodes) {
            cbp_code[c.value] = c.code;
            cbp_len[c.value] = c.len;
        }
        for (auto& c : efx::kMotionCodes) {
            mv_code[c.value + 16] = c.code;
            mv_len[c.value + 16] = c.len;
        }
        for (auto& c : efx::kTypePCodes) {
            type_p_code[c.value] = c.code;
            type_p_len[c.value] = c.len;
        }
        for (auto& c : kDctCodes) {
            dct_code[c.run][c.level] = c.code;
            dct_len[c.run][c.level] = c.len;
        }
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++)
                cosv[u][x] = (float)(std::cos((2 * x + 1) * u * M_PI / 16) * (u ? 0.5 : 0.5 / std::sqrt(2.0)));
        double s[8];
        s[0] = 1.0;
        for (int k = 1; k < 8; k++)
            s[k] = std::sqrt(2.0) * std::cos(k * M_PI / 16);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++)
                premul[i * 8 + j] = (uint8_t)std::floor(32.0 * s[i] * s[j] + 0.5);
    }
};
const Books& books()
{
    static Books b;
    return b;
}

// ---------------------------------------------------------------------------------------
// pictures (planar inside the generator; the decoder's strip layout is its own business)

struct Picture {
    std::vector<uint8_t> y, c0, c1;  // c0 = block 4 plane, c1 = block 5 plane
    Picture() : y(W * H), c0(CW * CH), c1(CW * CH) {}
};

// ---------------------------------------------------------------------------------------
// the decoder's reconstruction arithmetic, restated for the local decode loop

inline void idct_pass(int* b, int st, bool final)
{
    int i0 = b[0], i1 = b[st], i2 = b[2 * st], i3 = b[3 * st], i4 = b[4 * st], i5 = b[5 * st], i6 = b[6 * st], i7 = b[7 * st];
    int b3 = i2 + i6, b4 = i5 - i3, t1 = i1 + i7, t2 = i3 + i5, b6 = i1 - i7, b7 = t1 + t2;
    int x4 = ((b6 * 473 - b4 * 196 + 128) >> 8) - b7;
    int x0 = x4 - (((t1 - t2) * 362 + 128) >> 8);
    int x1 = i0 - i4;
    int x2 = (((i2 - i6) * 362 + 128) >> 8) - b3;
    int x3 = i0 + i4;
    int y3 = x1 + x2, y4 = x3 + b3, y5 = x1 - x2, y6 = x3 - b3;
    int y7 = -x0 - ((b4 * 473 + b6 * 196 + 128) >> 8);
    int o[8] = {b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7};
    for (int k = 0; k < 8; k++)
        b[k * st] = final ? (o[k] + 128) >> 8 : o[k];
}

inline int clamp248(int v) { return v < 0 ? 0 : (v > 248 ? 248 : v); }

// levels[64] in scan order (levels[0] = DC value for intra).  Writes/adds into dst (stride) when `commit`.
// Returns false when a value handed to the decoder's clamp leaves -256..511: the reference clamps through a
// 768-entry table (PIN, player.cpp:183-236) and reads whatever lies beside it for anything else, so such a
// block is outside the domain on which the reference is defined (FLAG_HUGE_LEVELS keeps its streams inside).
bool reconstruct(const int* levels, bool intra, int qscale, const uint8_t* qm, uint8_t* dst, int stride, bool commit = true)
{
    const Books& B = books();
    int coef[64] = {0};
    int last = -1;
    if (intra) {
        coef[0] = levels[0] << 8;
        last = 0;
    }
    for (int n = intra ? 1 : 0; n < 64; n++) {
        int v = levels[n];
        if (!v)
            continue;
        if (v == kCodedZero)
            v = 0;  // decoded like any other level (player.cpp:1110-1121): 0 -> +-1 -> ... -> odd
        int zz = kZigZag[n];
        v <<= 1;
        if (!intra)
            v += v < 0 ? -1 : 1;
        v = (v * qscale * qm[zz]) / 16;
        if ((v & 1) == 0)
            v -= v > 0 ? 1 : -1;
        v = v > 2047 ? 2047 : (v < -2048 ? -2048 : v);
        coef[zz] = v * B.premul[zz];
        last = n;
    }
    // the decoder's "n == 1" shortcut: exactly one coefficient, at scan position 0
    bool dc_only = last == 0 && (intra || levels[0] != 0);
    if (last < 0)
        return true;
    if (dc_only) {
        bool only_first = true;
        for (int n = 1; n < 64; n++)
            if (levels[n])
                only_first = false;
        if (only_first) {
            int dc = coef[0] >> 8;
            for (int y = 0; y < 8 && commit; y++)
                for (int x = 0; x < 8; x++)
                    dst[y * stride + x] = intra ? (uint8_t)dc : (uint8_t)clamp248(dc + dst[y * stride + x]);
            return true;  // dc + pixel lies in -256..503
        }
    }
    for (int c = 0; c < 8; c++)
        idct_pass(coef + c, 8, false);
    for (int r = 0; r < 8; r++)
        idct_pass(coef + r * 8, 1, true);
    bool in_range = true;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const int v = coef[y * 8 + x] + (intra ? 0 : dst[y * stride + x]);
            in_range &= v >= -256 && v <= 511;
            if (commit)
                dst[y * stride + x] = (uint8_t)clamp248(v);
        }
    return in_range;
}

// forward DCT of an 8x8 block of ints (orthonormal scaling: DC = 8 x mean)
void fdct(const int* in, float* out)
{
    const Books& B = books();
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int x = 0; x < 8; x++)
                s += in[y * 8 + x] * B.cosv[u][x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int y = 0; y < 8; y++)
                s += tmp[y * 8 + u] * B.cosv[v][y];
            out[v * 8 + u] = s;
        }
}

// ---------------------------------------------------------------------------------------
// stream encoder

struct Encoder {
    uint32_t k;
    uint32_t flags;
    Lcg rng;
    BitWriter bw;
    Picture cur, ref, src;
    uint8_t intra_q[64], non_intra_q[64];
    int qscale_base;
    int f_code, full_pel;
    int ox = 0, oy = 0;  // cumulative source offset
    std::vector<uint32_t> pic_offsets;

    explicit Encoder(uint32_t stream, uint32_t fl, int q_override = 0) : k(stream), flags(fl)
    {
        rng.s = 0xE5F10000u + stream;
        qscale_base = q_override ? q_override : 4 + (int)(stream & 7);
        if ((fl & FLAG_HUGE_LEVELS) && !q_override)
            qscale_base = 1;
        f_code = (stream & 1) ? 2 : 1;
        full_pel = (stream & 7) == 7;
        memcpy(intra_q, efx::kDefaultIntraQ, 64);
        memset(non_intra_q, 16, 64);
    }

    // source texture: sawtooth gradients + position-hashed grain, translating with (ox,oy)
    void make_source(int f)
    {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int u = x + ox, v = y + oy;
                uint32_t h = (uint32_t)(u * 73856093) ^ (uint32_t)(v * 19349663) ^ (k * 83492791u);
                h = h * 1664525u + 1013904223u;
                int t = (3 * u + 5 * v + 13 * (int)k) & 0xFF;
                int tri = t < 128 ? t : 256 - t;  // 0..128 triangle wave: smooth gradients
                int val = 40 + tri * 5 / 4 + (int)(h >> 28) + (int)(rng.next() >> 31);
                if ((flags & FLAG_FLAT_BRIGHT) && ((u >> 5) + (v >> 4)) % 3 == 0)
                    val = 255;
                else
                    val = std::min(std::max(val, 16), 235);
                if (flags & FLAG_HUGE_LEVELS) {
                    // per 8x8 source block: vertical / horizontal step, checkerboard, impulse, white noise, or the
                    // smooth texture above -- full range, so that at quantiser_scale 1 the levels leave +-127
                    uint32_t hb = (uint32_t)((u >> 3) * 2654435761u) ^ (uint32_t)((v >> 3) * 40503u) ^ (k * 97u);
                    hb = hb * 1664525u + 1013904223u;
                    switch (hb >> 27) {  // (6 blocks in 32)
                    case 0: val = (u & 4) ? 255 : 0; break;
                    case 1: val = (v & 4) ? 0 : 255; break;
                    case 2: val = ((u ^ v) & 1) ? 250 : 5; break;
                    case 3: val = ((u & 7) == 3 && (v & 7) == 4) ? 255 : 0; break;
                    case 4: val = (int)(h >> 24); break;
                    case 5: val = (int)(rng.next() >> 24); break;  // differs from picture to picture: big residuals
                    default: break;
                    }
                }
                src.y[y * W + x] = (uint8_t)val;
            }
        for (int y = 0; y < CH; y++)
            for (int x = 0; x < CW; x++) {
                int u = x + (ox >> 1), v = y + (oy >> 1);
                src.c0[y * CW + x] = (uint8_t)(128 + ((u + f + (int)k) & 31) - 16);
                src.c1[y * CW + x] = (uint8_t)(128 + ((v + 2 * f + (int)k) & 31) - 16);
            }
    }

    void sequence_header()
    {
        bw.start_code(0xB3);
        bw.put(W, 12);
        bw.put(H, 12);
        bw.put(1, 4);      // square pels
        bw.put(4, 4);      // 29.97 Hz (as splash_ts)
        bw.put(3750, 18);  // 1.5 Mbit/s in 400 bit/s units
        bw.put(1, 1);
        bw.put(20, 10);  // vbv buffer size
        bw.put(0, 1);
        if (flags & FLAG_CUSTOM_MATRICES) {
            // The decoder stores loaded matrices in arrival order and indexes them by raster
            // position (player.cpp:646-651,1113): whatever we send IS what it applies.
            for (int i = 0; i < 64; i++) {
                intra_q[i] = (uint8_t)(i == 0 ? 8 : 10 + ((i * 7 + k) % 40));
                non_intra_q[i] = (uint8_t)(12 + ((i * 5 + k) % 24));
            }
            bw.put(1, 1);
            for (int i = 0; i < 64; i++)
                bw.put(intra_q[i], 8);
            bw.put(1, 1);
            for (int i = 0; i < 64; i++)
                bw.put(non_intra_q[i], 8);
        } else {
            bw.put(0, 1);
            bw.put(0, 1);
        }
    }

    void gop_header(int f)
    {
        bw.start_code(0xB8);
        int pictures = f % 30, seconds = (f / 30) % 60;
        bw.put((seconds << 6) | pictures | (1 << 12), 25);  // marker bit inside time_code
        bw.put(1, 1);                                        // closed_gop
        bw.put(0, 1);
        bw.put(0, 5);
    }

    // FLAG_ODD_HEADERS: user_data / extension units the decoder must pass over (player.cpp:1328-1330).  The
    // reference does not skip a payload, it resumes its marker hunt INSIDE it (24 bits discarded, 8 bits taken as the
    // next marker, player.cpp:1360-1363): the user data here is 4-byte groups whose fourth byte is a marker value
    // without effect (a slice row beyond the picture, user data, extension, an unknown code) and an extension
    // carries zero bytes only, so the hunt arrives at the next real start code.
    void odd_units()
    {
        if (!(flags & FLAG_ODD_HEADERS))
            return;
        const uint32_t r = rng.next();
        if ((r >> 30) == 0) {
            bw.start_code(0xB2);
            static const uint8_t harmless[8] = {'!', 'x', 0xB2, 0xB5, 0xB9, 0xFF, 0x0E, 0xAF};
            const int groups = 1 + (int)((r >> 8) & 3);
            for (int g = 0; g < groups; g++) {
                bw.put('a' + ((r >> (g * 3)) & 15), 8);
                bw.put('A' + ((r >> (g * 2 + 4)) & 15), 8);
                bw.put('0' + ((r >> (g + 12)) & 7), 8);
                bw.put(harmless[(r >> (16 + 3 * g)) & 7], 8);
            }
        }
        if (((r >> 28) & 3) == 1) {
            bw.start_code(0xB5);
            for (int z = (int)((r >> 5) & 3); z > 0; z--)
                bw.put(0, 8);
        }
    }

    void picture_header(int temporal, int type, int hdr_type)
    {
        bw.start_code(0x00);
        bw.put(temporal, 10);
        if (hdr_type != type) {
            // a picture_coding_type the decoder ignores (player.cpp:710-717): it returns after these 13 bits and its
            // marker hunt must find nothing but zero bits up to the next start code
            bw.put(hdr_type, 3);
            bw.align();
            for (int z = temporal % 3; z > 0; z--)
                bw.put(0, 8);
            return;
        }
        bw.put(type, 3);
        bw.put(0xFFFF, 16);
        if (type == 2) {
            bw.put(full_pel, 1);
            bw.put(f_code, 3);
        }
        bw.put(0, 1);  // extra_bit_picture
    }

    void put_mba(int inc)
    {
        const Books& B = books();
        if ((flags & FLAG_LONG_SKIPS) && (rng.next() >> 29) == 0)
            bw.put(B.mba_code[34], B.mba_len[34]);  // macroblock_stuffing
        while (inc > 33) {
            bw.put(B.mba_code[35], B.mba_len[35]);
            inc -= 33;
        }
        bw.put(B.mba_code[inc], B.mba_len[inc]);
    }

    void put_dc(int diff, bool luma)
    {
        int a = std::abs(diff), size = 0;
        while (a >> size)
            size++;
        static const uint8_t ycode[9] = {4, 0, 1, 5, 6, 14, 30, 62, 126}, ylen[9] = {3, 2, 2, 3, 3, 4, 5, 6, 7};
        static const uint8_t ccode[9] = {0, 1, 2, 6, 14, 30, 62, 126, 254}, clen[9] = {2, 2, 2, 3, 4, 5, 6, 7, 8};
        if (luma)
            bw.put(ycode[size], ylen[size]);
        else
            bw.put(ccode[size], clen[size]);
        if (size)
            bw.put(diff > 0 ? diff : diff + (1 << size) - 1, size);
    }

    void put_coefs(const int* levels, bool intra)
    {
        const Books& B = books();
        int run = 0;
        bool first = !intra;
        for (int n = intra ? 1 : 0; n < 64; n++) {
            int v = levels[n];
            if (!v) {
                run++;
                continue;
            }
            if (flags & FLAG_HUGE_LEVELS) {
                // every form of the escape the decoder accepts (player.cpp:1092-1099): "xx" for -127..127, "00 xx" = 0..255,
                // "80 xx" = xx - 256 = -256..-1; a quarter of the levels that have a code are escaped as well
                const uint32_t r = rng.next();
                const int lv = v == kCodedZero ? 0 : v;
                const bool has_code = lv != 0 && std::abs(lv) <= 40 && run < 32 && (B.dct_len[run][std::abs(lv)] || (run == 0 && std::abs(lv) == 1));
                // No start code may appear inside a slice: 23 zero bits and a one (the code books alone guarantee that;
                // the long forms of small levels, which ISO 11172-2 forbids for this very reason, do not).  A code word
                // whose leading zeros would stretch the zeros already written to 22 is sent as an escape (5 zeros).
                int lz = 0;
                if (has_code && !(run == 0 && std::abs(lv) == 1)) {
                    const int len = B.dct_len[run][std::abs(lv)];
                    lz = len - (32 - __builtin_clz((uint32_t)B.dct_code[run][std::abs(lv)]));
                }
                if (!has_code || (r >> 30) == 0 || bw.zrun + lz >= 22) {
                    bw.put(efx::kDctEscapeCode, efx::kDctEscapeLen);
                    bw.put(run, 6);
                    const bool can_short = lv != 0 && lv >= -127 && lv <= 127;
                    if (can_short && ((r >> 28) & 1))
                        bw.put(lv & 0xFF, 8);
                    else if (lv >= 0) {
                        bw.put(0, 8);
                        bw.put(lv, 8);
                    } else {
                        bw.put(128, 8);
                        bw.put(lv + 256, 8);
                    }
                    first = false;
                    run = 0;
                    continue;
                }
            }
            int a = std::abs(v);
            if (run == 0 && a == 1) {
                if (first)
                    bw.put(2 | (v < 0), 2);  // "1s"
                else
                    bw.put(6 | (v < 0), 3);  // "11s"
            } else if (a <= 40 && run < 32 && B.dct_len[run][a]) {
                bw.put(B.dct_code[run][a], B.dct_len[run][a]);
                bw.put(v < 0, 1);
            } else {
                bw.put(efx::kDctEscapeCode, efx::kDctEscapeLen);
                bw.put(run, 6);
                if (a < 128)
                    bw.put(v & 0xFF, 8);
                else if (v > 0) {
                    bw.put(0, 8);
                    bw.put(v, 8);
                } else {
                    bw.put(128, 8);
                    bw.put(v + 256, 8);
                }
            }
            first = false;
            run = 0;
        }
        bw.put(2, 2);  // end_of_block
    }

    // quantise one block; returns true if any level is non-zero (intra: always true)
    bool quantise(const int* pix, bool intra, int qscale, int* levels, bool inject = true)
    {
        float F[64];
        fdct(pix, F);
        bool any = false;
        const uint8_t* qm = intra ? intra_q : non_intra_q;
        for (int n = 0; n < 64; n++) {
            int zz = kZigZag[n];
            int l;
            if (intra && n == 0)
                l = std::min(std::max((int)std::lround(F[0] / 8), 0), 255);
            else if (intra)
                l = (int)std::lround(F[zz] * 8 / (qscale * qm[zz]));
            else
                l = (int)(F[zz] * 8 / (qscale * qm[zz]));  // dead zone
            if (!(intra && n == 0))
                l = std::min(std::max(l, -255), 255);
            if ((flags & FLAG_HUGE_LEVELS) && inject && !(intra && n == 0)) {
                const uint32_t r = rng.next();
                if (l <= -250 && (r >> 31))
                    l = -256;  // "80 00"
                else if ((r >> 20) == 1)
                    l = (int)((r >> 8) & 0x1FF) - 256;  // a full-range level anywhere (also behind runs > 31), one in 4096
            }
            levels[n] = l;
            any |= l != 0;
        }
        if ((flags & FLAG_HUGE_LEVELS) && inject) {
            // a coefficient coded with level 0 ("00 00": 16 zero bits) only behind the last one, where end_of_block ("10")
            // follows: anything else could complete a start code
            const uint32_t r = rng.next();
            int last = intra ? 0 : -1;
            for (int n = 0; n < 64; n++)
                if (levels[n] && !(intra && n == 0))
                    last = n;
            if ((r >> 28) == 0 && last < 63) {
                levels[last + 1 + (int)((r >> 8) % (uint32_t)(63 - last))] = kCodedZero;
                any = true;
            }
        }
        return any || intra;
    }

    // quantise(), and for FLAG_HUGE_LEVELS make sure the decoder's clamp stays inside its table (see reconstruct):
    // first without the injected levels, then with the AC levels halved until the block fits
    bool quantise_in_domain(const int* pix, bool intra, int qscale, int* levels, const uint8_t* qm, uint8_t* dp, int dst_st)
    {
        bool any = quantise(pix, intra, qscale, levels);
        if (!(flags & FLAG_HUGE_LEVELS) || reconstruct(levels, intra, qscale, qm, dp, dst_st, false))
            return any;
        any = quantise(pix, intra, qscale, levels, false);
        while (!reconstruct(levels, intra, qscale, qm, dp, dst_st, false)) {
            any = intra;
            for (int n = intra ? 1 : 0; n < 64; n++) {
                levels[n] /= 2;
                any |= levels[n] != 0;
            }
        }
        return any;
    }

    static void plane_ptr(Picture& p, int blk, int mbx, int mby, uint8_t*& ptr, int& stride)
    {
        if (blk < 4) {
            ptr = &p.y[(mby * 16 + (blk >> 1) * 8) * W + mbx * 16 + (blk & 1) * 8];
            stride = W;
        } else {
            ptr = &(blk == 4 ? p.c0 : p.c1)[(mby * 8) * CW + mbx * 8];
            stride = CW;
        }
    }

    // half-pel prediction into cur (exactly the decoder's four cases)
    void predict_plane(const std::vector<uint8_t>& r, std::vector<uint8_t>& d, int pw, int px, int py, int size, int mbx, int mby)
    {
        int hx = px & 1, hy = py & 1, x0 = px >> 1, y0 = py >> 1;
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++) {
                const uint8_t* s = &r[(y0 + y) * pw + x0 + x];
                int v;
                if (!hx && !hy)
                    v = s[0];
                else if (hx && !hy)
                    v = (s[0] + s[1] + 1) >> 1;
                else if (!hx)
                    v = (s[0] + s[pw] + 1) >> 1;
                else
                    v = (s[0] + s[1] + s[pw] + s[pw + 1] + 2) >> 2;
                d[(mby * size + y) * pw + mbx * size + x] = (uint8_t)v;
            }
    }

    void predict_mb(int mbx, int mby, int h, int v)  // h,v: half-pel luma vector after full_pel scaling
    {
        int x = (mbx << 5) + h, y = (mby << 5) + v;
        predict_plane(ref.y, cur.y, W, x, y, 16, mbx, mby);
        x >>= 1;
        y >>= 1;
        predict_plane(ref.c0, cur.c0, CW, x, y, 8, mbx, mby);
        predict_plane(ref.c1, cur.c1, CW, x, y, 8, mbx, mby);
    }

    void put_motion(int delta)  // delta already wrapped into the f_code range
    {
        const Books& B = books();
        int r = f_code - 1;
        if (delta == 0 || r == 0) {
            bw.put(B.mv_code[delta + 16], B.mv_len[delta + 16]);
            return;
        }
        int a = std::abs(delta) - 1;
        int code = (a >> r) + 1;
        if (delta < 0)
            code = -code;
        bw.put(B.mv_code[code + 16], B.mv_len[code + 16]);
        bw.put(a & ((1 << r) - 1), r);
    }

    void encode_picture(int f, int type, int hdr_type, int dxg, int dyg)
    {
        const Books& B = books();
        picture_header(f, type, hdr_type);
        odd_units();
        // slice layout: start rows of the slices
        std::vector<int> starts;
        if (flags & FLAG_WIDE_SLICES) {
            const int s5[5] = {0, 2, 5, 7, 10};  // like the embedded clips (codes 1,3,6,8,11)
            starts.assign(s5, s5 + 5);
        } else
            for (int r = 0; r < MBH; r++)
                starts.push_back(r);
        for (size_t si = 0; si < starts.size(); si++) {
            int row0 = starts[si], row1 = si + 1 < starts.size() ? starts[si + 1] : MBH;
            int first_mb = row0 * MBW, last_mb = row1 * MBW - 1;
            int qscale = qscale_base;
            if (si && (flags & FLAG_ODD_HEADERS) && (rng.next() >> 29) == 0)
                odd_units();
            bw.start_code(row0 + 1);
            bw.put(qscale, 5);
            if (flags & FLAG_SLICE_EXTRA) {
                // extra_information_slice: ISO 11172-2 reserves it, the reference skips it a byte at a time
                const uint32_t r = rng.next();
                for (int e = (int)(r >> 30); e > 0; e--) {
                    bw.put(1, 1);  // extra_bit_slice
                    bw.put(0x80 | ((r >> (7 * e)) & 0x7F), 8);
                }
            }
            bw.put(0, 1);  // extra_bit_slice
            int dc_pred[3] = {128, 128, 128};
            int pmv_h = 0, pmv_v = 0;  // in coded units (full pels when full_pel)
            int pending_skip = 0;
            for (int mb = first_mb; mb <= last_mb; mb++) {
                int mbx = mb % MBW, mby = mb / MBW;
                uint32_t roll = rng.next() >> 16;
                int decision;  // 0 intra, 1 mc+coded, 2 mc only, 3 skipped
                if (type == 1)
                    decision = 0;
                else {
                    int pct = (int)(roll % 100);
                    decision = pct < 10 ? 0 : (pct < 70 ? 1 : (pct < 85 ? 2 : 3));
                    if (flags & FLAG_LONG_SKIPS)
                        decision = (mb > first_mb + 1 && mb < last_mb - 1 && (mb % 44) != 0) ? 3 : decision;
                    if (decision == 3 && (mb == first_mb || mb == last_mb))
                        decision = 2;
                }
                if (decision == 3) {
                    // skipped: decoder copies the co-located macroblock and resets predictors
                    predict_mb(mbx, mby, 0, 0);
                    pending_skip++;
                    continue;
                }
                int inc = pending_skip + 1;
                if (mb == first_mb)
                    inc = 1;
                if (pending_skip) {
                    dc_pred[0] = dc_pred[1] = dc_pred[2] = 128;
                    pmv_h = pmv_v = 0;
                    pending_skip = 0;
                }
                put_mba(inc);

                // occasional quantiser change
                int new_q = qscale;
                if ((roll & 0x1F00) == 0x1F00)
                    new_q = std::min(std::max(qscale_base + (int)((roll >> 5) & 3) - 1, 1), 31);
                if ((flags & FLAG_HUGE_LEVELS) && (roll & 0x700) == 0x300) {
                    static const int jumps[4] = {1, 2, 31, 9};  // 31: the +-2047 / -2048 clamp of player.cpp:1117-1120
                    new_q = jumps[(roll >> 5) & 3];
                }
                bool quant = new_q != qscale;

                if (decision == 0) {  // intra macroblock
                    if (type == 1) {
                        if (quant)
                            bw.put(1, 2), bw.put(new_q, 5);
                        else
                            bw.put(1, 1);
                    } else {
                        int t = quant ? 17 : 1;
                        bw.put(B.type_p_code[t], B.type_p_len[t]);
                        if (quant)
                            bw.put(new_q, 5);
                    }
                    qscale = new_q;
                    pmv_h = pmv_v = 0;
                    for (int blk = 0; blk < 6; blk++) {
                        uint8_t *sp, *dp;
                        int sst, dst_st;
                        plane_ptr(src, blk, mbx, mby, sp, sst);
                        plane_ptr(cur, blk, mbx, mby, dp, dst_st);
                        int pix[64], levels[64];
                        for (int y = 0; y < 8; y++)
                            for (int x = 0; x < 8; x++)
                                pix[y * 8 + x] = sp[y * sst + x];
                        quantise_in_domain(pix, true, qscale, levels, intra_q, dp, dst_st);
                        int comp = blk < 4 ? 0 : blk - 3;
                        put_dc(levels[0] - dc_pred[comp], blk < 4);
                        dc_pred[comp] = levels[0];
                        put_coefs(levels, true);
                        reconstruct(levels, true, qscale, intra_q, dp, dst_st);
                    }
                    continue;
                }

                // inter macroblock: vector = global motion + jitter, kept inside the picture
                dc_pred[0] = dc_pred[1] = dc_pred[2] = 128;
                int unit = full_pel ? 2 : 1;  // half-pels per coded unit
                int range = 16 << (f_code - 1);  // coded units: [-range, range-1]
                int jit = f_code == 2 ? 5 : 3;
                int h = 2 * dxg, v = 2 * dyg;
                if ((roll & 3) == 0) {  // a quarter of the macroblocks get a perturbed vector
                    h += (int)(rng.next() >> 24) % jit - jit / 2;
                    v += (int)(rng.next() >> 24) % jit - jit / 2;
                }
                h = (h / unit);
                v = (v / unit);
                h = std::min(std::max(h, -range), range - 1);
                v = std::min(std::max(v, -range), range - 1);
                // clip so that the (size+1)^2 luma fetch stays inside the picture
                auto clip = [&](int mv, int mbpos, int limit) {
                    int lo = -(mbpos * 32), hi = (limit - 16) * 2 - mbpos * 32;  // half-pel bounds
                    int hp = mv * unit;
                    hp = std::min(std::max(hp, lo), hi);
                    int r = hp / unit;
                    if (r * unit < lo)
                        r++;
                    if (r * unit > hi)
                        r--;
                    return r;
                };
                h = clip(h, mbx, W);
                v = clip(v, mby, H);
                predict_mb(mbx, mby, h * unit, v * unit);

                int levels[6][64];
                int cbp = 0;
                if (decision == 1) {
                    for (int blk = 0; blk < 6; blk++) {
                        uint8_t *sp, *dp;
                        int sst, dst_st;
                        plane_ptr(src, blk, mbx, mby, sp, sst);
                        plane_ptr(cur, blk, mbx, mby, dp, dst_st);
                        int pix[64];
                        for (int y = 0; y < 8; y++)
                            for (int x = 0; x < 8; x++)
                                pix[y * 8 + x] = (int)sp[y * sst + x] - (int)dp[y * dst_st + x];
                        if (quantise_in_domain(pix, false, new_q, levels[blk], non_intra_q, dp, dst_st))
                            cbp |= 0x20 >> blk;
                    }
                }
                bool has_mv = h != 0 || v != 0 || cbp == 0;  // "no mc, not coded" does not exist
                if (!cbp)
                    quant = false, new_q = qscale;
                int t = (cbp ? 2 : 0) | (has_mv ? 8 : 0) | (quant ? 16 : 0);
                if (quant && !has_mv)
                    t = 18;
                bw.put(B.type_p_code[t], B.type_p_len[t]);
                if (quant)
                    bw.put(new_q, 5);
                qscale = new_q;
                if (has_mv) {
                    auto wrap = [&](int d) {
                        if (d < -range)
                            d += 2 * range;
                        else if (d > range - 1)
                            d -= 2 * range;
                        return d;
                    };
                    put_motion(wrap(h - pmv_h));
                    put_motion(wrap(v - pmv_v));
                    pmv_h = h;
                    pmv_v = v;
                } else
                    pmv_h = pmv_v = 0;
                if (cbp) {
                    bw.put(B.cbp_code[cbp], B.cbp_len[cbp]);
                    for (int blk = 0; blk < 6; blk++)
                        if (cbp & (0x20 >> blk)) {
                            uint8_t* dp;
                            int dst_st;
                            plane_ptr(cur, blk, mbx, mby, dp, dst_st);
                            put_coefs(levels[blk], false);
                            reconstruct(levels[blk], false, qscale, non_intra_q, dp, dst_st);
                        }
                }
            }
        }
    }

    void run(int n_pictures, int gop)
    {
        for (int f = 0; f < n_pictures; f++) {
            int type = ((flags & FLAG_I_ONLY) || (f % gop) == 0) ? 1 : 2;
            int dxg = (2 * f + (int)k) % 5, dyg = f % 3;  // source translation this frame (pixels)
            if (f) {
                ox += dxg;
                oy += dyg;
            }
            make_source(f);
            bw.align();
            pic_offsets.push_back((uint32_t)bw.buf.size());
            int hdr_type = type;
            if ((flags & FLAG_ODD_HEADERS) && type == 2) {
                const int g = f % gop;
                if (g >= 2 && g % 3 == 2) {
                    static const int odd[4] = {3, 4, 0, 7};  // B, D, forbidden, reserved
                    hdr_type = odd[(g / 3 + (int)k) & 3];    // coded as P with the last real P header's f_code
                } else {
                    f_code = 1 + (int)((k + (uint32_t)f) % 3 == 0);
                    full_pel = (k + (uint32_t)f) % 5 == 0;
                }
            }
            if ((f % gop) == 0) {
                sequence_header();
                odd_units();
                gop_header(f);
                odd_units();
            }
            // the picture content moved by (+dxg,+dyg) in source coordinates, i.e. the matching
            // reference block lies at (+dxg,+dyg) relative to the current block
            encode_picture(f % 1024, type, hdr_type, type == 2 ? dxg : 0, type == 2 ? dyg : 0);
            std::swap(cur, ref);
        }
        bw.align();
        pic_offsets.push_back((uint32_t)bw.buf.size());
    }
};

// ---------------------------------------------------------------------------------------
// transport-stream wrapper: PID 0x100, one PES (stream id E0) with PTS per picture

void ts_wrap(const uint8_t* es, const std::vector<uint32_t>& offs, std::vector<uint8_t>& out)
{
    int cc = 0;
    for (size_t p = 0; p + 1 < offs.size(); p++) {
        std::vector<uint8_t> pes;
        int64_t pts = 129003 + 3003 * (int64_t)p;
        const uint8_t hdr[9] = {0, 0, 1, 0xE0, 0, 0, 0x80, 0x80, 5};
        pes.insert(pes.end(), hdr, hdr + 9);
        pes.push_back((uint8_t)(0x21 | ((pts >> 29) & 0x0E)));
        pes.push_back((uint8_t)(pts >> 22));
        pes.push_back((uint8_t)(0x01 | ((pts >> 14) & 0xFE)));
        pes.push_back((uint8_t)(pts >> 7));
        pes.push_back((uint8_t)(0x01 | ((pts << 1) & 0xFE)));
        pes.insert(pes.end(), es + offs[p], es + offs[p + 1]);
        size_t pos = 0;
        bool first = true;
        while (pos < pes.size()) {
            size_t left = pes.size() - pos;
            uint8_t pkt[188];
            pkt[0] = 0x47;
            pkt[1] = (uint8_t)((first ? 0x40 : 0) | 0x01);
            pkt[2] = 0x00;
            size_t n;
            if (left >= 184) {
                pkt[3] = (uint8_t)(0x10 | (cc & 15));
                n = 184;
                memcpy(pkt + 4, &pes[pos], n);
            } else {
                pkt[3] = (uint8_t)(0x30 | (cc & 15));
                size_t stuff = 184 - left;  // adaptation_field_length byte + (stuff-1) bytes
                pkt[4] = (uint8_t)(stuff - 1);
                if (stuff > 1) {
                    pkt[5] = 0;
                    memset(pkt + 6, 0xFF, stuff - 2);
                }
                n = left;
                memcpy(pkt + 4 + stuff, &pes[pos], n);
            }
            out.insert(out.end(), pkt, pkt + 188);
            pos += n;
            cc++;
            first = false;
        }
    }
}

struct Batch {
    std::vector<std::vector<uint8_t>> es;
    std::vector<std::vector<uint32_t>> offs;
};

}  // namespace

extern "C" {

// Generate `n_streams` streams (ids first_id ...) of n_pictures pictures each on `threads` host
// threads.  Returns an opaque handle.
void* efxgen_batch_create(uint32_t first_id, int n_streams, int n_pictures, int gop, uint32_t flags, int threads)
{
    if (n_streams <= 0 || n_pictures <= 0 || gop <= 0)
        return nullptr;
    Batch* b = new Batch;
    b->es.resize(n_streams);
    b->offs.resize(n_streams);
    books();
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n_streams)
                break;
            if (flags & FLAG_RATE_1500K) {
                // quantiser_scale whose stream size is closest to the target: bisection for the largest
                // scale still at or above it (size falls with the scale), then the neighbour is tried
                const size_t target = (size_t)6250 * n_pictures;
                auto size_at = [&](int q) {
                    Encoder t(first_id + (uint32_t)i, flags, q);
                    t.run(n_pictures, gop);
                    return t.bw.buf.size();
                };
                int lo = 1, hi = 31;
                while (lo < hi) {
                    int mid = (lo + hi + 1) / 2;
                    if (size_at(mid) >= target)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                if (lo < 31) {
                    const size_t a = size_at(lo), b2 = size_at(lo + 1);
                    if (a > target && b2 < target && target - b2 < a - target)
                        lo++;
                }
                Encoder e(first_id + (uint32_t)i, flags, lo);
                e.run(n_pictures, gop);
                b->es[i].swap(e.bw.buf);
                b->offs[i].swap(e.pic_offsets);
                continue;
            }
            Encoder e(first_id + (uint32_t)i, flags);
            e.run(n_pictures, gop);
            b->es[i].swap(e.bw.buf);
            b->offs[i].swap(e.pic_offsets);
        }
    };
    threads = std::max(1, std::min(threads, n_streams));
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++)
        pool.emplace_back(work);
    work();
    for (auto& t : pool)
        t.join();
    return b;
}

void efxgen_batch_destroy(void* h) { delete (Batch*)h; }

// FNV-1a-64 of a byte range (basis cbf29ce484222325, prime 100000001b3), for callers that check bulk
// output against the committed golden hashes without a byte loop in Python
uint64_t efxgen_fnv1a64(const uint8_t* p, uint64_t n, uint64_t h)
{
    for (uint64_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

uint64_t efxgen_batch_es_size(void* h, int i)
{
    Batch* b = (Batch*)h;
    return (i < 0 || i >= (int)b->es.size()) ? 0 : b->es[i].size();
}

// copies stream i's ES; returns bytes copied (0 if cap too small)
uint64_t efxgen_batch_es_copy(void* h, int i, uint8_t* out, uint64_t cap)
{
    Batch* b = (Batch*)h;
    if (i < 0 || i >= (int)b->es.size() || cap < b->es[i].size())
        return 0;
    memcpy(out, b->es[i].data(), b->es[i].size());
    return b->es[i].size();
}

// picture access-unit offsets of stream i (n_pictures + 1 entries)
int efxgen_batch_offsets(void* h, int i, uint32_t* out, int cap)
{
    Batch* b = (Batch*)h;
    if (i < 0 || i >= (int)b->offs.size())
        return -1;
    int n = (int)b->offs[i].size();
    for (int j = 0; j < n && j < cap; j++)
        out[j] = b->offs[i][j];
    return n;
}

// TS-wrapped form of stream i (PID 0x100, one PES + PTS per picture).  Call with out == NULL
// to get the size.
uint64_t efxgen_batch_ts(void* h, int i, uint8_t* out, uint64_t cap)
{
    Batch* b = (Batch*)h;
    if (i < 0 || i >= (int)b->es.size())
        return 0;
    std::vector<uint8_t> ts;
    ts_wrap(b->es[i].data(), b->offs[i], ts);
    if (out && cap >= ts.size())
        memcpy(out, ts.data(), ts.size());
    return ts.size();
}

}  // extern "C"
