/* efx_oracle.c -- CPU restatement of the espflix hot path: MPEG-1 decode, TS demux (video + audio),
 * composite video incl. overlay / slide, PDM, SBC audio decode, trick-play index.
 *
 * TEST INFRASTRUCTURE ONLY (see efx_oracle.h).  Plain C99, serial, written for clarity.
 * Every function cites the reference file:line (under /root/reference) it restates.
 * The restatement is deliberately structured differently from both the reference and the
 * HIP product (bit-serial VLC tries built from code strings, pixel-at-a-time prediction) so
 * that agreement between the three is meaningful.
 *
 * Defined behaviour where the reference has undefined behaviour (never reached by valid
 * streams; the HIP path defines the same):
 *   - motion vectors pointing outside the picture: source coordinates are clamped;
 *   - reconstructed values outside the PIN table domain [-256,511]: plain clamp to 0..248;
 *   - a slice running past the last macroblock row stops;
 *   - an invalid VLC code ends the slice;
 *   - pictures must be 352x192 (the frame store is fixed size, video.h:30-34).
 */
#define _DEFAULT_SOURCE /* M_PI */
#include "efx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FB_WIDTH 352
#define FB_HEIGHT 192
#define FB_STRIDE 528
#define STRIP_ROWS 16
#define STRIPS 12
#define STRIP_BYTES (FB_STRIDE * STRIP_ROWS)

/* ------------------------------------------------------------------------------------ */
/* hashing                                                                              */

uint64_t efxo_fnv1a64(const uint8_t* p, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

/* ------------------------------------------------------------------------------------ */
/* Frame store: 12 strips of 16 rows x 528 bytes; a row is 352 luma bytes followed by 176
 * chroma bytes; strip rows 0-7 carry the "cr" plane, rows 8-15 the "cb" plane
 * (video.h:36-44, player.cpp:25-46).                                                   */

typedef struct {
    uint8_t* strip[STRIPS];
} frame_t;

static void frame_init(frame_t* f) /* player.cpp:25-31 (zero filled, slack for over-read) */
{
    for (int i = 0; i < STRIPS; i++)
        f->strip[i] = (uint8_t*)calloc(STRIP_BYTES + 16, 1);
}

static void frame_free(frame_t* f)
{
    for (int i = 0; i < STRIPS; i++)
        free(f->strip[i]);
}

static uint8_t* luma_row(const frame_t* f, int y) /* Frame::get_y, player.cpp:33-36 */
{
    return f->strip[y >> 4] + (y & 15) * FB_STRIDE;
}

static uint8_t* cr_row(const frame_t* f, int c) /* Frame::get_cr, player.cpp:38-41 */
{
    return f->strip[c >> 3] + (c & 7) * FB_STRIDE + FB_WIDTH;
}

static uint8_t* cb_row(const frame_t* f, int c) /* Frame::get_cb, player.cpp:43-46 */
{
    return f->strip[c >> 3] + ((c & 7) + 8) * FB_STRIDE + FB_WIDTH;
}

/* ------------------------------------------------------------------------------------ */
/* VLC code books (ISO/IEC 11172-2 annex B) as "code=value" strings, turned into binary  */
/* tries at start-up.  The reference keeps the same books as packed tree tables          */
/* (player.cpp:59-116) and a hand-unrolled prefix decoder for the DCT book (532-644).    */

static const char* const book_mba = /* table B-1; 34 = stuffing, 35 = escape */
    "1=1 011=2 010=3 0011=4 0010=5 00011=6 00010=7 0000111=8 0000110=9 00001011=10 00001010=11 "
    "00001001=12 00001000=13 00000111=14 00000110=15 0000010111=16 0000010110=17 0000010101=18 "
    "0000010100=19 0000010011=20 0000010010=21 00000100011=22 00000100010=23 00000100001=24 "
    "00000100000=25 00000011111=26 00000011110=27 00000011101=28 00000011100=29 00000011011=30 "
    "00000011010=31 00000011001=32 00000011000=33 00000001111=34 00000001000=35";

static const char* const book_type_i = "1=1 01=17"; /* B-2a: intra, intra+quant */

static const char* const book_type_p = /* B-2b: bit0 intra, bit1 pattern, bit3 forward, bit4 quant */
    "1=10 01=2 001=8 00011=1 00010=26 00001=18 000001=17";

static const char* const book_cbp = /* table B-3 */
    "111=60 1101=4 1100=8 1011=16 1010=32 10011=12 10010=48 10001=20 10000=40 01111=28 01110=44 "
    "01101=52 01100=56 01011=1 01010=61 01001=2 01000=62 001111=24 001110=36 001101=3 001100=63 "
    "0010111=5 0010110=9 0010101=17 0010100=33 0010011=6 0010010=10 0010001=18 0010000=34 "
    "00011111=7 00011110=11 00011101=19 00011100=35 00011011=13 00011010=49 00011001=21 "
    "00011000=41 00010111=14 00010110=50 00010101=22 00010100=42 00010011=15 00010010=51 "
    "00010001=23 00010000=43 00001111=25 00001110=37 00001101=26 00001100=38 00001011=29 "
    "00001010=45 00001001=53 00001000=57 00000111=30 00000110=46 00000101=54 00000100=58 "
    "000000111=31 000000110=47 000000101=55 000000100=59 000000011=27 000000010=39";

static const char* const book_motion = /* table B-4 */
    "1=0 010=1 011=-1 0010=2 0011=-2 00010=3 00011=-3 0000110=4 0000111=-4 00001010=5 "
    "00001011=-5 00001000=6 00001001=-6 00000110=7 00000111=-7 0000010110=8 0000010111=-8 "
    "0000010100=9 0000010101=-9 0000010010=10 0000010011=-10 00000100010=11 00000100011=-11 "
    "00000100000=12 00000100001=-12 00000011110=13 00000011111=-13 00000011100=14 "
    "00000011101=-14 00000011010=15 00000011011=-15 00000011000=16 00000011001=-16";

/* DCT coefficients, table B-5c/d-f without the leading-1 codes (handled by the caller like
 * player.cpp:1073-1087) and with escape = -1.  value = run*256 + level. */
static const char* const book_dct =
    "011=1.1 0100=0.2 0101=2.1 00101=0.3 00110=4.1 00111=3.1 000100=7.1 000101=6.1 000110=1.2 "
    "000111=5.1 000001=E 0000100=2.2 0000101=9.1 0000110=0.4 0000111=8.1 00100000=13.1 "
    "00100001=0.6 00100010=12.1 00100011=11.1 00100100=3.2 00100101=1.3 00100110=0.5 "
    "00100111=10.1 0000001000=16.1 0000001001=5.2 0000001010=0.7 0000001011=2.3 0000001100=1.4 "
    "0000001101=15.1 0000001110=14.1 0000001111=4.2 "
    "000000010000=0.11 000000010001=8.2 000000010010=4.3 000000010011=0.10 000000010100=2.4 "
    "000000010101=7.2 000000010110=21.1 000000010111=20.1 000000011000=0.9 000000011001=19.1 "
    "000000011010=18.1 000000011011=1.5 000000011100=3.3 000000011101=0.8 000000011110=6.2 "
    "000000011111=17.1 "
    "0000000010000=10.2 0000000010001=9.2 0000000010010=5.3 0000000010011=3.4 0000000010100=2.5 "
    "0000000010101=1.7 0000000010110=1.6 0000000010111=0.15 0000000011000=0.14 "
    "0000000011001=0.13 0000000011010=0.12 0000000011011=26.1 0000000011100=25.1 "
    "0000000011101=24.1 0000000011110=23.1 0000000011111=22.1 "
    "00000000010000=0.31 00000000010001=0.30 00000000010010=0.29 00000000010011=0.28 "
    "00000000010100=0.27 00000000010101=0.26 00000000010110=0.25 00000000010111=0.24 "
    "00000000011000=0.23 00000000011001=0.22 00000000011010=0.21 00000000011011=0.20 "
    "00000000011100=0.19 00000000011101=0.18 00000000011110=0.17 00000000011111=0.16 "
    "000000000010000=0.40 000000000010001=0.39 000000000010010=0.38 000000000010011=0.37 "
    "000000000010100=0.36 000000000010101=0.35 000000000010110=0.34 000000000010111=0.33 "
    "000000000011000=0.32 000000000011001=1.14 000000000011010=1.13 000000000011011=1.12 "
    "000000000011100=1.11 000000000011101=1.10 000000000011110=1.9 000000000011111=1.8 "
    "0000000000010000=1.18 0000000000010001=1.17 0000000000010010=1.16 0000000000010011=1.15 "
    "0000000000010100=6.3 0000000000010101=16.2 0000000000010110=15.2 0000000000010111=14.2 "
    "0000000000011000=13.2 0000000000011001=12.2 0000000000011010=11.2 0000000000011011=31.1 "
    "0000000000011100=30.1 0000000000011101=29.1 0000000000011110=28.1 0000000000011111=27.1";

#define TRIE_MAX 512
#define TRIE_NONE (-32768)
typedef struct {
    int16_t child[TRIE_MAX][2]; /* 0 = absent */
    int16_t value[TRIE_MAX];    /* TRIE_NONE on interior nodes */
    int nodes;
} trie_t;

static void trie_build(trie_t* t, const char* book)
{
    memset(t, 0, sizeof(*t));
    for (int i = 0; i < TRIE_MAX; i++)
        t->value[i] = TRIE_NONE;
    t->nodes = 1;
    const char* p = book;
    while (*p) {
        while (*p == ' ')
            p++;
        if (!*p)
            break;
        int node = 0;
        while (*p == '0' || *p == '1') {
            int b = *p++ - '0';
            if (!t->child[node][b])
                t->child[node][b] = (int16_t)t->nodes++;
            node = t->child[node][b];
        }
        p++; /* '=' */
        int v;
        if (*p == 'E') {
            v = -1;
            p++;
        } else {
            int sign = 1;
            if (*p == '-') {
                sign = -1;
                p++;
            }
            v = 0;
            while (*p >= '0' && *p <= '9')
                v = v * 10 + (*p++ - '0');
            if (*p == '.') { /* run.level */
                p++;
                int l = 0;
                while (*p >= '0' && *p <= '9')
                    l = l * 10 + (*p++ - '0');
                v = v * 256 + l;
            }
            v *= sign;
        }
        t->value[node] = (int16_t)v;
    }
}

static trie_t T_mba, T_type_i, T_type_p, T_cbp, T_motion, T_dct;
static int tries_ready = 0;

static void tries_init(void)
{
    if (tries_ready)
        return;
    trie_build(&T_mba, book_mba);
    trie_build(&T_type_i, book_type_i);
    trie_build(&T_type_p, book_type_p);
    trie_build(&T_cbp, book_cbp);
    trie_build(&T_motion, book_motion);
    trie_build(&T_dct, book_dct);
    tries_ready = 1;
}

/* ------------------------------------------------------------------------------------ */
/* constant tables                                                                      */

static const uint8_t zigzag[64] = { /* scan order, ISO 11172-2 fig. 2-D.30 (player.cpp:150-159) */
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t intra_default[64] = { /* default intra matrix, raster (player.cpp:172-181) */
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34,
    34, 38, 22, 22, 26, 27, 29, 34, 37, 40, 22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32,
    35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

/* IDCT pre-multipliers: round(32 * s_i * s_j), s_0 = 1, s_k = sqrt(2) cos(k pi/16)
 * (player.cpp:161-170; derived here, compared with the reference's table in the tests). */
static uint8_t premul[64];

static void premul_init(void)
{
    double s[8];
    s[0] = 1.0;
    for (int k = 1; k < 8; k++)
        s[k] = sqrt(2.0) * cos(k * M_PI / 16);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            premul[i * 8 + j] = (uint8_t)floor(32.0 * s[i] * s[j] + 0.5);
}

/* exported for the table-pin test */
void efxo_tables(uint8_t zz_out[64], uint8_t premul_out[64])
{
    premul_init();
    memcpy(zz_out, zigzag, 64);
    memcpy(premul_out, premul, 64);
}

/* ------------------------------------------------------------------------------------ */
/* decoder state                                                                        */

typedef struct {
    /* byte source */
    const uint8_t* data;
    size_t len;
    int format;
    size_t ts_next;     /* offset of the next TS packet to consume */
    const uint8_t* cur; /* current payload run */
    const uint8_t* end;
    int eos_stage; /* 0 stream, 1 serving the eos pad, 2 exhausted */
    uint8_t* es_sink;  /* optional: collect the bytes handed to the bit reader (ts_to_es) */
    size_t es_cap, es_len;
    size_t es_bytes; /* elementary-stream bytes handed to the bit reader so far (trace only) */
    int64_t pts, last_pts;

    /* bit window: low `cnt` bits of `win` are unread, MSB first */
    uint64_t win;
    int cnt;

    /* sequence / picture state (player.h:96-141) */
    int mb_width, mb_height;
    int pic_type, full_pel, r_size, qscale;
    int mb_x, mb_y;
    int dc_y, dc_cr, dc_cb;
    int mv_h, mv_v;
    uint8_t intra_q[64], non_intra_q[64];

    frame_t fb[2];
    int fb_index;
    frame_t* ref;
    frame_t* curf;
    int bad_size;
    int custom_q; /* the sequence header in force loaded a quantiser matrix (trace only) */
    int ended;

    /* results */
    long pushed;
    long pictures;
    uint8_t* frames_out;
    int64_t* pts_out;
    uint64_t* hash_out;
    long max_frames;
} dec_t;

/* optional parse trace (tests/parse_harness.cpp): every slice, macroblock, coefficient and block
 * result in decode order, see efx_oracle.h */
static efxo_trace_fn trace_fn;
static void* trace_user;
void efxo_set_trace(efxo_trace_fn fn, void* user)
{
    trace_fn = fn;
    trace_user = user;
}
#define TRACE(kind, a, b, c, e) \
    do { \
        if (trace_fn) \
            trace_fn(trace_user, kind, a, b, c, e); \
    } while (0)

static const uint8_t eos_pad[8] = {0, 0, 1, 0xB7, 0, 0, 1, 0xB7}; /* player.cpp:456 */

static int64_t parse_pts(const uint8_t* d, int flags) /* player.cpp:299-307 */
{
    flags = (flags >> 2) & 0x30;
    if ((d[0] & 0xF0) != flags)
        return -1;
    int64_t n = ((int64_t)(d[0] & 0x0E)) << 29;
    n += (int64_t)((((d[1] << 8) | d[2]) >> 1) << 15);
    return n + ((((d[3] << 8) | d[4])) >> 1);
}

/* Next byte for the bit reader: MpegDecoder::more()+demux() (player.cpp:381-436,459-493). */
static int src_byte(dec_t* d)
{
    if (d->cur < d->end)
        return *d->cur++;
    if (d->eos_stage) { /* pad consumed: the reference would block for ever */
        d->eos_stage = 2;
        return 0;
    }
    if (d->format == EFXO_FMT_ES) {
        /* treat the ES as one payload run; at its end behave like a zero-length Buffer */
        if (d->cur == NULL && d->len) {
            d->cur = d->data + 1;
            d->end = d->data + d->len;
            return d->data[0];
        }
        d->eos_stage = 1;
        d->cur = eos_pad;
        d->end = eos_pad + 8;
        return 0; /* player.cpp:472 hands back a literal 0 before the pad */
    }
    for (;;) {
        if (d->ts_next + 188 > d->len) { /* read() returned 0 -> Buffer.len == 0 (player.cpp:469) */
            d->eos_stage = 1;
            d->cur = eos_pad;
            d->end = eos_pad + 8;
            return 0;
        }
        const uint8_t* p = d->data + d->ts_next;
        d->ts_next += 188;
        if (p[0] != 0x47)
            return 0; /* "ts lost sync", player.cpp:477-480 */
        int pid = ((p[1] << 8) + p[2]) & 0x1fff;
        const uint8_t* pay = p + 4;
        if (p[3] & 0x20)
            pay = p + 5 + p[4];
        if (!(p[3] & 0x10))
            continue;
        const uint8_t* pend = p + 188;
        int64_t pts = -1;
        if (p[1] & 0x40) { /* payload_unit_start: fixed-offset PES header (player.cpp:387-406) */
            /* The reference reads the nine header bytes wherever they fall; when they do not fit
             * in the packet that is a read outside it, whose result depends on how the transport
             * was buffered, not on the stream.  Defined here: such a packet is dropped. */
            if (pay + 9 > pend)
                continue;
            const uint8_t* q = pay + 6;
            int flags = (q[0] << 8) | q[1];
            pay = q + 3 + q[2];
            q += 3;
            if ((flags & 0x0080) && q + 5 <= pend)
                pts = parse_pts(q, flags);
        }
        if (pid != 0x100)
            continue; /* audio 0x101/0x102 goes to push_audio, everything else is dropped */
        if (pts != -1)
            d->pts = pts;
        /* player.cpp:419 returns *_data++ without checking for an empty payload: with nothing
         * left in the packet it would hand the bit reader the byte AFTER the packet (the next
         * packet's sync byte, or whatever follows the 8-packet Buffer).  Defined here: a packet
         * without payload bytes contributes nothing. */
        if (pay >= pend)
            continue;
        d->cur = pay + 1;
        d->end = pend;
        return *pay;
    }
}

static int next_byte(dec_t* d)
{
    int b = src_byte(d);
    if (d->eos_stage == 0)
        d->es_bytes++;
    if (d->es_sink && d->eos_stage == 0) {
        if (d->es_len < d->es_cap)
            d->es_sink[d->es_len] = (uint8_t)b;
        d->es_len++;
    }
    return b;
}

/* FILL_BITS (player.cpp:348-352): top up byte-wise while fewer than 24 bits are buffered */
static void fill(dec_t* d)
{
    while (d->cnt < 24) {
        d->win = (d->win << 8) | (uint64_t)next_byte(d);
        d->cnt += 8;
    }
}

static void need(dec_t* d, int n) /* only get_bits(25) in gop() can ask for more than fill() gives */
{
    while (d->cnt < n) {
        d->win = (d->win << 8) | (uint64_t)next_byte(d);
        d->cnt += 8;
    }
}

static int peek_bits(dec_t* d, int n) /* player.cpp:502-507 */
{
    fill(d);
    need(d, n);
    return (int)((d->win >> (d->cnt - n)) & ((1u << n) - 1));
}

static int get_bits(dec_t* d, int n) /* player.cpp:509-514 */
{
    int v = peek_bits(d, n);
    d->cnt -= n;
    return v;
}

static int raw_bit(dec_t* d) /* the un-refilled "(_b >> --_b_count) & 1" reads, player.cpp:1101 */
{
    need(d, 1);
    d->cnt--;
    return (int)((d->win >> d->cnt) & 1);
}

static int get_vlc(dec_t* d, const trie_t* t, int* ok) /* player.cpp:516-530 */
{
    fill(d);
    int node = 0;
    for (;;) {
        need(d, 1);
        d->cnt--;
        int b = (int)((d->win >> d->cnt) & 1);
        node = t->child[node][b];
        if (!node) {
            *ok = 0;
            return 0;
        }
        if (t->value[node] != TRIE_NONE)
            return t->value[node];
    }
}

/* ------------------------------------------------------------------------------------ */
/* headers (player.cpp:646-730)                                                         */

static void sequence_header(dec_t* d) /* player.cpp:658-678 */
{
    int w = get_bits(d, 12);
    int h = get_bits(d, 12);
    get_bits(d, 4);  /* pel aspect */
    get_bits(d, 4);  /* picture rate */
    get_bits(d, 18); /* bit rate */
    get_bits(d, 12); /* marker, vbv buffer size, constrained flag */
    d->custom_q = 0;
    if (get_bits(d, 1)) {
        /* stored in arrival (zig-zag) order and later indexed by raster position: kept as is */
        for (int i = 0; i < 64; i++)
            d->intra_q[i] = (uint8_t)get_bits(d, 8);
        d->custom_q = 1;
    } else
        memcpy(d->intra_q, intra_default, 64);
    if (get_bits(d, 1)) {
        for (int i = 0; i < 64; i++)
            d->non_intra_q[i] = (uint8_t)get_bits(d, 8);
        d->custom_q = 1;
    } else
        memset(d->non_intra_q, 16, 64);
    d->mb_width = (w + 15) >> 4;
    d->mb_height = (h + 15) >> 4;
    d->bad_size = (w != FB_WIDTH || h != FB_HEIGHT);
}

static void push_frame(dec_t* d, const frame_t* f, int64_t pts)
{
    if (d->pushed < d->max_frames) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (int s = 0; s < STRIPS; s++) {
            h = efxo_fnv1a64(f->strip[s], STRIP_BYTES, h);
            if (d->frames_out)
                memcpy(d->frames_out + (size_t)d->pushed * EFXO_FRAME_BYTES + (size_t)s * STRIP_BYTES, f->strip[s],
                       STRIP_BYTES);
        }
        if (d->hash_out)
            d->hash_out[d->pushed] = h;
        if (d->pts_out)
            d->pts_out[d->pushed] = pts;
    }
    d->pushed++;
}

static void flush_picture(dec_t* d, int mode) /* player.cpp:692-702 */
{
    if (d->last_pts != -1 || mode) {
        push_frame(d, &d->fb[d->fb_index & 1], d->last_pts);
        d->ref = &d->fb[d->fb_index++ & 1];
        d->curf = &d->fb[d->fb_index & 1];
    }
    if (!mode)
        d->last_pts = d->pts;
}

static void picture_header(dec_t* d) /* player.cpp:704-724 */
{
    if (d->format == EFXO_FMT_ES)
        d->pts = d->pictures; /* ES mode: synthetic pts = picture index (see efx_oracle.h) */
    d->pictures++;
    flush_picture(d, 0);
    get_bits(d, 10); /* temporal reference */
    d->pic_type = get_bits(d, 3);
    if (d->pic_type != 1 && d->pic_type != 2)
        return; /* B/D headers are ignored; their slices still decode with the P book */
    get_bits(d, 16); /* vbv delay */
    if (d->pic_type == 2) {
        d->full_pel = get_bits(d, 1);
        d->r_size = get_bits(d, 3) - 1;
    }
}

/* ------------------------------------------------------------------------------------ */
/* prediction (player.cpp:732-889)                                                      */

static int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

static int ref_pixel(const frame_t* f, int plane, int x, int y)
{
    if (plane == 0)
        return luma_row(f, clampi(y, 0, FB_HEIGHT - 1))[clampi(x, 0, FB_WIDTH - 1)];
    y = clampi(y, 0, FB_HEIGHT / 2 - 1);
    x = clampi(x, 0, FB_WIDTH / 2 - 1);
    return plane == 1 ? cr_row(f, y)[x] : cb_row(f, y)[x];
}

static uint8_t* cur_pixel_row(const frame_t* f, int plane, int y)
{
    return plane == 0 ? luma_row(f, y) : (plane == 1 ? cr_row(f, y) : cb_row(f, y));
}

/* mocomp (player.cpp:733-821): pos_* are half-pel positions inside the plane; the four
 * cases are full, horizontal, vertical and diagonal half-pel with upward rounding. */
static void motion_comp(dec_t* d, int plane, int pos_x, int pos_y, int size)
{
    int hx = pos_x & 1, hy = pos_y & 1;
    int x0 = pos_x >> 1, y0 = pos_y >> 1;
    int dst_x = d->mb_x * size;
    int dst_y = d->mb_y * size;
    for (int y = 0; y < size; y++) {
        uint8_t* dst = cur_pixel_row(d->curf, plane, dst_y + y) + dst_x;
        for (int x = 0; x < size; x++) {
            int a = ref_pixel(d->ref, plane, x0 + x, y0 + y);
            int v;
            if (!hx && !hy)
                v = a;
            else if (hx && !hy)
                v = (a + ref_pixel(d->ref, plane, x0 + x + 1, y0 + y) + 1) >> 1;
            else if (!hx)
                v = (a + ref_pixel(d->ref, plane, x0 + x, y0 + y + 1) + 1) >> 1;
            else
                v = (a + ref_pixel(d->ref, plane, x0 + x + 1, y0 + y) + ref_pixel(d->ref, plane, x0 + x, y0 + y + 1) +
                     ref_pixel(d->ref, plane, x0 + x + 1, y0 + y + 1) + 2) >> 2;
            dst[x] = (uint8_t)v;
        }
    }
}

static void predict_zero(dec_t* d) /* player.cpp:861-867: straight copy of the co-located MB */
{
    for (int y = 0; y < 16; y++)
        memcpy(luma_row(d->curf, d->mb_y * 16 + y) + d->mb_x * 16, luma_row(d->ref, d->mb_y * 16 + y) + d->mb_x * 16, 16);
    for (int y = 0; y < 8; y++) {
        memcpy(cr_row(d->curf, d->mb_y * 8 + y) + d->mb_x * 8, cr_row(d->ref, d->mb_y * 8 + y) + d->mb_x * 8, 8);
        memcpy(cb_row(d->curf, d->mb_y * 8 + y) + d->mb_x * 8, cb_row(d->ref, d->mb_y * 8 + y) + d->mb_x * 8, 8);
    }
}

static void predict(dec_t* d) /* player.cpp:870-889 */
{
    int h = d->mv_h, v = d->mv_v;
    if (h == 0 && v == 0) {
        predict_zero(d);
        return;
    }
    if (d->full_pel) {
        h <<= 1;
        v <<= 1;
    }
    int x = (d->mb_x << 5) + h;
    int y = (d->mb_y << 5) + v;
    motion_comp(d, 0, x, y, 16);
    x >>= 1; /* chroma uses the floor of the halved POSITION (not of the vector) */
    y >>= 1;
    motion_comp(d, 1, x, y, 8);
    motion_comp(d, 2, x, y, 8);
}

static int motion_vector(dec_t* d, int m, int r_size, int* ok) /* player.cpp:891-910 */
{
    int scale = 1 << r_size;
    int code = get_vlc(d, &T_motion, ok);
    int delta;
    if (code != 0 && scale != 1) {
        delta = ((abs(code) - 1) << r_size) + get_bits(d, r_size) + 1;
        if (code < 0)
            delta = -delta;
    } else
        delta = code;
    m += delta;
    if (m > (scale << 4) - 1)
        m -= scale << 5;
    else if (m < -(scale << 4))
        m += scale << 5;
    return m;
}

/* ------------------------------------------------------------------------------------ */
/* block layer (player.cpp:922-1236)                                                    */

/* One 8-point pass of the scaled integer IDCT (player.cpp:938-995).  in/out stride `st`;
 * `final` adds the output rounding of the row pass. */
static void idct_1d(int* b, int st, int final)
{
    int i0 = b[0], i1 = b[st], i2 = b[2 * st], i3 = b[3 * st], i4 = b[4 * st], i5 = b[5 * st], i6 = b[6 * st],
        i7 = b[7 * st];
    int b1 = i4;
    int b3 = i2 + i6;
    int b4 = i5 - i3;
    int t1 = i1 + i7;
    int t2 = i3 + i5;
    int b6 = i1 - i7;
    int b7 = t1 + t2;
    int x4 = ((b6 * 473 - b4 * 196 + 128) >> 8) - b7;
    int x0 = x4 - (((t1 - t2) * 362 + 128) >> 8);
    int x1 = i0 - b1;
    int x2 = (((i2 - i6) * 362 + 128) >> 8) - b3;
    int x3 = i0 + b1;
    int y3 = x1 + x2;
    int y4 = x3 + b3;
    int y5 = x1 - x2;
    int y6 = x3 - b3;
    int y7 = -x0 - ((b4 * 473 + b6 * 196 + 128) >> 8);
    int o[8] = {b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7};
    for (int k = 0; k < 8; k++)
        b[k * st] = final ? (o[k] + 128) >> 8 : o[k];
}

static int pin(int v) /* PIN(), player.cpp:183-236: clamp to 0..0xF8 */
{
    return v < 0 ? 0 : (v > 0xF8 ? 0xF8 : v);
}

/* returns 0, or -1 when the coefficient index overruns (player.cpp:1106-1107, block dropped),
 * or -2 on an invalid code */
static int decode_block(dec_t* d, int blk, int intra)
{
    int b[64];
    memset(b, 0, sizeof(b));
    const uint8_t* q = d->non_intra_q;
    int n = 0;

    if (intra) { /* DC size + differential, player.cpp:1010-1068 */
        int size;
        int pb = peek_bits(d, 10);
        if (blk < 4) { /* luminance, table B-5a */
            b[0] = d->dc_y;
            pb >>= 1;
            if (!(pb & 0x100)) {
                size = 1 + (pb >> 7);
                d->cnt -= 2;
            } else if (!(pb & 0x80)) {
                size = (pb & 0x40) ? 3 : 0;
                d->cnt -= 3;
            } else {
                size = 4;
                pb <<= 2;
                while (pb & 0x100) {
                    pb <<= 1;
                    size++;
                }
                d->cnt -= size - 1;
            }
        } else { /* chrominance, table B-5b */
            b[0] = (blk == 4) ? d->dc_cr : d->dc_cb;
            if (!(pb & 0x200)) {
                size = pb >> 8;
                d->cnt -= 2;
            } else {
                size = 1;
                do {
                    pb <<= 1;
                    size++;
                } while (pb & 0x200);
                d->cnt -= size < 10 ? size : 10;
            }
        }
        if (size) {
            int delta = get_bits(d, size);
            if (delta & (1 << (size - 1)))
                b[0] += delta;
            else
                b[0] += (int)((~0u << size) | (unsigned)(delta + 1));
            if (blk == 4)
                d->dc_cr = b[0];
            else if (blk == 5)
                d->dc_cb = b[0];
            else
                d->dc_y = b[0];
        }
        TRACE(EFXO_T_COEF, blk, 0, b[0], 0);
        b[0] = (int)((unsigned)b[0] << 8);
        q = d->intra_q;
        n = 1;
    }
    int count = n;

    for (;;) { /* run/level pairs, player.cpp:1070-1122 */
        int p = peek_bits(d, 2);
        if (n && p == 2) {
            d->cnt -= 2; /* end of block */
            break;
        }
        int run, v;
        if (p >> 1) { /* "1s" first coefficient / "11s" later: run 0, level 1 */
            d->cnt -= n ? 2 : 1;
            run = 0;
            v = raw_bit(d) ? -1 : 1;
        } else {
            int ok = 1;
            int rl = get_vlc(d, &T_dct, &ok);
            if (!ok) {
                TRACE(EFXO_T_BLOCK, blk, -2, count, 0);
                return -2;
            }
            if (rl == -1) { /* escape: 6 bit run, 8 or 16 bit level (player.cpp:1092-1099) */
                run = get_bits(d, 6);
                v = get_bits(d, 8);
                int form = v == 0 ? 1 : (v == 128 ? 2 : 0);
                if (v == 0)
                    v = get_bits(d, 8);
                else if (v == 128)
                    v = get_bits(d, 8) - 256;
                else if (v > 128)
                    v -= 256;
                TRACE(EFXO_T_ESCAPE, form, run, v, 0);
            } else {
                run = rl >> 8;
                v = rl & 0xFF;
                if (raw_bit(d))
                    v = -v;
            }
        }
        n += run;
        if (n >= 64) {
            TRACE(EFXO_T_BLOCK, blk, -1, count, 0);
            return -1;
        }
        TRACE(EFXO_T_COEF, blk, n, v, 0);
        count++;
        int zz = zigzag[n++];

        /* reconstruction, player.cpp:1110-1121 */
        v <<= 1;
        if (!intra)
            v += (v < 0) ? -1 : 1;
        v = (v * d->qscale * q[zz]) / 16; /* truncation toward zero */
        if ((v & 1) == 0)
            v -= (v > 0) ? 1 : -1; /* v == 0 becomes +1 */
        if (v > 2047)
            v = 2047;
        else if (v < -2048)
            v = -2048;
        b[zz] = v * premul[zz];
    }

    TRACE(EFXO_T_BLOCK, blk, 0, count, 0);
    /* destination, player.cpp:1124-1131 */
    int plane = blk < 4 ? 0 : blk - 3;
    int px, py;
    if (blk < 4) {
        px = d->mb_x * 16 + (blk & 1) * 8;
        py = d->mb_y * 16 + (blk >> 1) * 8;
    } else {
        px = d->mb_x * 8;
        py = d->mb_y * 8;
    }

    if (n == 1) { /* single coefficient at index 0, player.cpp:1133-1140 */
        int dc = b[0] >> 8;
        if (intra) { /* copy_block_dc (player.cpp:1175-1187): replicated WITHOUT clamping */
            uint32_t w = (uint32_t)dc;
            w |= w << 8;
            w |= w << 16;
            for (int y = 0; y < 8; y++) {
                uint8_t* dst = cur_pixel_row(d->curf, plane, py + y) + px;
                for (int x = 0; x < 8; x++)
                    dst[x] = (uint8_t)(w >> (8 * (x & 3)));
            }
        } else { /* add_block_dc, player.cpp:1214-1236 */
            for (int y = 0; y < 8; y++) {
                uint8_t* dst = cur_pixel_row(d->curf, plane, py + y) + px;
                for (int x = 0; x < 8; x++)
                    dst[x] = (uint8_t)pin(dc + dst[x]);
            }
        }
        return 0;
    }

    for (int c = 0; c < 8; c++)
        idct_1d(b + c, 8, 0);
    for (int r = 0; r < 8; r++)
        idct_1d(b + r * 8, 1, 1);

    for (int y = 0; y < 8; y++) { /* copy_block / add_block, player.cpp:1151-1212 */
        uint8_t* dst = cur_pixel_row(d->curf, plane, py + y) + px;
        for (int x = 0; x < 8; x++)
            dst[x] = (uint8_t)pin(b[y * 8 + x] + (intra ? 0 : dst[x]));
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* slice / macroblock layer (player.cpp:823-833,1238-1316)                              */

static void reset_predictors(dec_t* d) /* player.cpp:726-730 */
{
    d->dc_y = d->dc_cr = d->dc_cb = 128;
    d->mv_h = d->mv_v = 0;
}

static int advance_mb(dec_t* d) /* inc_mb, player.cpp:823-833 (its argument is ignored there) */
{
    d->mb_x++;
    while (d->mb_x >= d->mb_width) {
        d->mb_x -= d->mb_width;
        d->mb_y++;
    }
    return d->mb_y < d->mb_height;
}

static void decode_slice(dec_t* d, int code)
{
    d->mb_y = code - 2;
    d->mb_x = d->mb_width - 1; /* first advance wraps to column 0 of row code-1 */
    {
        int rejected = d->mb_y >= d->mb_height || d->bad_size || d->mb_width == 0;
        if (d->eos_stage == 0) {
            size_t bit = d->es_bytes * 8 - (size_t)d->cnt; /* of the first bit after the marker the hunt stopped at */
            TRACE(EFXO_T_SLICE_AT, (int)(bit >> 3), (int)(bit & 7), 0, 0);
        }
        TRACE(EFXO_T_SLICE, (int)d->pictures - 1, code,
              d->pic_type | (d->full_pel << 4) | (d->r_size << 8) | (rejected ? 0 : 1 << 16), d->custom_q);
    }
    if (d->mb_y >= d->mb_height || d->bad_size || d->mb_width == 0)
        return;
    reset_predictors(d);
    d->qscale = get_bits(d, 5);
    {
        int extra = 0, last = 0; /* extra_bit_slice / extra_information_slice, player.cpp:1261-1262 */
        while (get_bits(d, 1)) {
            last = (int)get_bits(d, 8);
            extra++;
        }
        if (extra)
            TRACE(EFXO_T_SLICE_EXTRA, extra, last, d->pic_type, 0);
    }

    for (int mb = 0;; mb++) {
        if (peek_bits(d, 23) == 0) /* slice_done, player.cpp:1238-1249 */
            break;
        if (d->eos_stage == 2)
            break;
        int ok = 1;
        int inc = 0;
        int i = get_vlc(d, &T_mba, &ok);
        while (ok && i == 34)
            i = get_vlc(d, &T_mba, &ok);
        while (ok && i == 35) {
            inc += 33;
            i = get_vlc(d, &T_mba, &ok);
        }
        if (!ok)
            return;
        inc += i;

        if (mb == 0) {
            if (!advance_mb(d))
                return;
        } else {
            if (inc > 1)
                reset_predictors(d);
            while (inc > 1) {
                if (!advance_mb(d))
                    return;
                predict_zero(d); /* skipped macroblocks copy the reference */
                TRACE(EFXO_T_MB, d->mb_y * d->mb_width + d->mb_x, 2, 0, 0);
                inc--;
            }
            if (!advance_mb(d))
                return;
        }

        int type = get_vlc(d, d->pic_type == 1 ? &T_type_i : &T_type_p, &ok);
        if (!ok)
            return;
        int intra = type & 1;
        if (type & 0x10)
            d->qscale = get_bits(d, 5);
        if (intra)
            d->mv_h = d->mv_v = 0;
        else {
            d->dc_y = d->dc_cr = d->dc_cb = 128;
            if (type & 0x08) { /* motion_vectors, player.cpp:912-920 */
                d->mv_h = motion_vector(d, d->mv_h, d->r_size, &ok);
                d->mv_v = motion_vector(d, d->mv_v, d->r_size, &ok);
                if (!ok)
                    return;
            } else
                d->mv_h = d->mv_v = 0;
            predict(d);
        }
        int cbp = (type & 2) ? get_vlc(d, &T_cbp, &ok) : (intra ? 63 : 0);
        if (!ok)
            return;
        TRACE(EFXO_T_MB, d->mb_y * d->mb_width + d->mb_x, intra | (d->qscale << 2), d->full_pel ? d->mv_h << 1 : d->mv_h,
              d->full_pel ? d->mv_v << 1 : d->mv_v);
        for (int blk = 0; blk < 6; blk++)
            if (cbp & (0x20 >> blk))
                if (decode_block(d, blk, intra) == -2)
                    return;
    }
}

/* ------------------------------------------------------------------------------------ */
/* top level: MpegDecoder::run()/marker() (player.cpp:1318-1367)                         */

static void run_decoder(dec_t* d)
{
    while (!d->ended && d->eos_stage != 2) {
        while (peek_bits(d, 24) == 0 && d->eos_stage != 2)
            get_bits(d, 1);
        get_bits(d, 24);
        int m = get_bits(d, 8);
        if (d->eos_stage == 2 && m != 0xB7)
            break; /* ran off the end-of-stream pad: the reference would block in _full_q.pop() */
        switch (m) {
        case 0xB3: sequence_header(d); break;
        case 0xB8: /* gop, player.cpp:680-690: 25 + 7 bits, values unused */
            get_bits(d, 25);
            get_bits(d, 7);
            break;
        case 0x00: picture_header(d); break;
        case 0xB7: d->ended = 1; break; /* pause(): end of this run */
        case 0xB2:
        case 0xB5: break;
        default:
            if (m >= 0x01 && m <= 0xAF)
                decode_slice(d, m);
            break;
        }
    }
}

static void dec_init(dec_t* d, const uint8_t* data, size_t len, int format)
{
    memset(d, 0, sizeof(*d));
    tries_init();
    premul_init();
    d->data = data;
    d->len = len;
    d->format = format;
    d->pts = d->last_pts = -1;
    frame_init(&d->fb[0]);
    frame_init(&d->fb[1]);
    d->fb_index = 0; /* MpegDecoder ctor, player.cpp:354-361 */
    d->ref = &d->fb[d->fb_index++ & 1];
    d->curf = &d->fb[d->fb_index & 1];
    memcpy(d->intra_q, intra_default, 64);
    memset(d->non_intra_q, 16, 64);
}

long efxo_decode(const uint8_t* data, size_t len, int format, int flush_last, uint8_t* frames_out, int64_t* pts_out,
                 uint64_t* hash_out, long max_frames)
{
    if (!data || (format != EFXO_FMT_ES && format != EFXO_FMT_TS))
        return -1;
    dec_t* d = (dec_t*)malloc(sizeof(dec_t));
    dec_init(d, data, len, format);
    d->frames_out = frames_out;
    d->pts_out = pts_out;
    d->hash_out = hash_out;
    d->max_frames = max_frames;
    run_decoder(d);
    if (flush_last)
        flush_picture(d, 1);
    long n = d->pushed;
    frame_free(&d->fb[0]);
    frame_free(&d->fb[1]);
    free(d);
    return n;
}

size_t efxo_ts_to_es(const uint8_t* ts, size_t len, uint8_t* es_out, size_t es_cap)
{
    dec_t* d = (dec_t*)malloc(sizeof(dec_t));
    dec_init(d, ts, len, EFXO_FMT_TS);
    d->es_sink = es_out;
    d->es_cap = es_cap;
    while (d->eos_stage == 0)
        next_byte(d);
    size_t n = d->es_len;
    frame_free(&d->fb[0]);
    frame_free(&d->fb[1]);
    free(d);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* composite video (video.cpp:514-934,1122-1198)                                        */

typedef struct {
    int pal;
    int line_width, line_count, samples_per_cc;
    float sample_rate;
    int hsync, hsync_long, hsync_short, burst_start, burst_width, active_start;
    int16_t burst0[64], burst1[64];
    uint32_t color_tab[768];
} video_t;

static uint32_t ire(double x) /* IRE(), video.cpp:520 */
{
    return ((uint32_t)((x + 40) * 255 / 3.3 / 147.5)) << 8;
}
#define SYNC_LEVEL ire(-40)
#define BLANKING_LEVEL ire(0)
#define BLACK_LEVEL ire(7.5)

static int usec(const video_t* v, float us) /* video.cpp:554-558 */
{
    uint32_t r = (uint32_t)(us * v->sample_rate);
    return (int)(((r + v->samples_per_cc) / (v->samples_per_cc << 1)) * (v->samples_per_cc << 1));
}

static int rup(float v) /* RUP(float), espflix.cpp:1071-1077: the argument is narrowed to float */
{
    if (v < 0)
        return -rup(-v);
    return (int)(v + 0.5);
}

/* One colour LUT entry: four carrier-phase samples of an 8-bit chroma value, centred on
 * 2*black, clipped to 0..127 and stored in the blitter's 0,2,1,3 byte order
 * (gen_palettes/swaz/pin, espflix.cpp:1079-1161). */
static uint32_t chroma_entry(int c, int use_cos, int negate)
{
    int black = (int)(BLACK_LEVEL >> 8);
    float scale = (float)black / 33;
    int amp = 128 - c;
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
        double w = use_cos ? cos(2 * M_PI * i / 4) : sin(2 * M_PI * i / 4);
        if (negate)
            w = -w;
        int p = rup(w * amp * scale) + 2 * black;
        p = p < 0 ? 0 : (p < 127 ? p : 127);
        v = (v << 8) | (uint32_t)p;
    }
    return (v & 0xFF0000FFu) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u);
}

static void video_setup(video_t* v, int ntsc) /* video_init + pal_init, video.cpp:572-630 */
{
    memset(v, 0, sizeof(*v));
    v->samples_per_cc = 4;
    if (ntsc) {
        v->sample_rate = 315.0 / 88 * v->samples_per_cc;
        v->line_width = 228 * v->samples_per_cc;
        v->line_count = 262;
        v->hsync_long = usec(v, 63.555 - 4.7);
        v->active_start = usec(v, 10);
        v->hsync = usec(v, 4.7);
        for (int c = 0; c < 256; c++) { /* uv_tab u, v, v (video.cpp:584-586) */
            v->color_tab[c] = chroma_entry(c, 0, 0);
            v->color_tab[256 + c] = v->color_tab[512 + c] = chroma_entry(c, 1, 0);
        }
    } else {
        int cc_width = 4;
        v->pal = 1;
        v->sample_rate = 4433618.75 * cc_width / 1000000.0;
        v->line_width = 284 * cc_width;
        v->line_count = 312;
        v->hsync_short = usec(v, 2);
        v->hsync_long = usec(v, 30);
        v->hsync = usec(v, 4.7);
        v->burst_start = usec(v, 5.6);
        v->burst_width = (int)(10 * cc_width + 4) & 0xFFFE;
        v->active_start = usec(v, 10.4);
        float phase = 2 * M_PI / 2;
        for (int i = 0; i < v->burst_width; i++) { /* video.cpp:621-629 */
            v->burst0[i] = (int16_t)(BLANKING_LEVEL + sin(phase + 3 * M_PI / 4) * BLANKING_LEVEL / 1.5);
            v->burst1[i] = (int16_t)(BLANKING_LEVEL + sin(phase - 3 * M_PI / 4) * BLANKING_LEVEL / 1.5);
            phase += 2 * M_PI / cc_width;
        }
        for (int c = 0; c < 256; c++) { /* sin_u, cos_v, cos_v_neg (video.cpp:589-591) */
            v->color_tab[c] = chroma_entry(c, 0, 0);
            v->color_tab[256 + c] = chroma_entry(c, 1, 0);
            v->color_tab[512 + c] = chroma_entry(c, 1, 1);
        }
    }
}

void efxo_video_params(int ntsc, int32_t out8[8])
{
    video_t v;
    video_setup(&v, ntsc);
    out8[0] = v.line_width;
    out8[1] = v.line_count;
    out8[2] = v.hsync;
    out8[3] = v.hsync_long;
    out8[4] = v.hsync_short;
    out8[5] = v.burst_start;
    out8[6] = v.burst_width;
    out8[7] = v.active_start;
}

void efxo_color_tab(int ntsc, uint32_t out768[768])
{
    video_t v;
    video_setup(&v, ntsc);
    memcpy(out768, v.color_tab, sizeof(v.color_tab));
}

static const uint32_t dither_tab[8] = { /* video.cpp:673-683: 4 lines x 2 frame phases */
    0x00020301, 0x03010002, 0x02030100, 0x01000203, 0x03010002, 0x00020301, 0x01000203, 0x02030100};

static uint32_t ld32(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static void st32(uint16_t* dst, uint32_t w) /* one 32-bit store = two little-endian samples */
{
    dst[0] = (uint16_t)w;
    dst[1] = (uint16_t)(w >> 16);
}

/* blit (video.cpp:690-804): `width` luma pixels from column x (x and width multiples of 8), four
 * luma pixels per step. */
static void blit_line(const video_t* v, const frame_t* f, uint16_t* dst, int line, int frame_counter, int x, int width)
{
    x &= ~3;
    const uint8_t* y = luma_row(f, line) + x;
    const uint8_t* u = cr_row(f, line >> 1) + (x >> 1);
    const uint8_t* w = cb_row(f, line >> 1) + (x >> 1);
    const uint8_t* u2 = u;
    const uint8_t* w2 = w;
    int odd = line & 1;
    if (odd) {
        int n = (line >> 1) + (line == 191 ? 0 : 1);
        u2 = cr_row(f, n) + (x >> 1);
        w2 = cb_row(f, n) + (x >> 1);
    }
    const uint32_t* tab_u = v->color_tab;
    const uint32_t* tab_v = v->color_tab + (odd ? 512 : 256);
    if (v->pal)
        dst += 80;
    uint32_t dither = dither_tab[(line & 3) + ((frame_counter & 1) << 2)];
    uint8_t lum = 0;
    for (int g = 0; g < width / 4; g++) {
        uint32_t u4 = ld32(u + (g >> 1) * 4), v4 = ld32(w + (g >> 1) * 4);
        if (odd) {
            u4 = ((u4 >> 1) & 0x7F7F7F7Fu) + ((ld32(u2 + (g >> 1) * 4) >> 1) & 0x7F7F7F7Fu);
            v4 = ((v4 >> 1) & 0x7F7F7F7Fu) + ((ld32(w2 + (g >> 1) * 4) >> 1) & 0x7F7F7F7Fu);
        }
        int sh = (g & 1) * 16;
        uint32_t p0 = (ld32(y + g * 4) + dither) & 0xFCFCFCFCu;
        uint32_t p1 = ((p0 >> 1) + (p0 >> 9)) & 0xFCFCFCFCu;
        p0 >>= 2;
        p1 >>= 2;
        uint32_t c = ((tab_u[(uint8_t)(u4 >> sh)] + tab_v[(uint8_t)(v4 >> sh)]) & 0xFCFCFCFCu) >> 2;
        lum = (uint8_t)(((uint8_t)p0 + lum) >> 1);
        st32(dst + 0, (((uint32_t)lum << 24) | ((p0 & 0xFF) << 8)) + c);
        st32(dst + 2, ((p1 << 24) | (p0 & 0xFF00)) + (c << 8));
        c = ((tab_u[(uint8_t)(u4 >> (sh + 8))] + tab_v[(uint8_t)(v4 >> (sh + 8))]) & 0xFCFCFCFCu) >> 2;
        st32(dst + 4, ((p1 << 16) | (p0 >> 8)) + c);
        st32(dst + 6, (((p1 << 8) & 0xFF000000u) | (p0 >> 16)) + (c << 8));
        lum = (uint8_t)(p0 >> 24);
        dst += 8;
    }
}

static void put_sync(uint16_t* line, int n) /* sync(), video.cpp:889-893 */
{
    for (int i = 0; i < n; i++)
        line[i] = (uint16_t)SYNC_LEVEL;
}

static void put_burst(const video_t* v, uint16_t* line, int line_counter) /* burst/burst_pal */
{
    if (v->pal) { /* video.cpp:636-644; _line_counter has already been advanced */
        const int16_t* b = (line_counter & 1) ? v->burst0 : v->burst1;
        for (int i = 0; i < v->burst_width; i++)
            line[v->burst_start + (i ^ 1)] = (uint16_t)b[i];
        return;
    }
    for (int i = v->hsync; i < v->hsync + 40; i += 4) { /* video.cpp:817-822 */
        line[i + 1] = (uint16_t)BLANKING_LEVEL;
        line[i + 0] = (uint16_t)(BLANKING_LEVEL + BLANKING_LEVEL / 2);
        line[i + 3] = (uint16_t)BLANKING_LEVEL;
        line[i + 2] = (uint16_t)(BLANKING_LEVEL - BLANKING_LEVEL / 2);
    }
}

static void put_blanking(const video_t* v, uint16_t* line, int vbl, int line_counter) /* video.cpp:904-914 */
{
    int sw = vbl ? v->hsync_long : v->hsync;
    put_sync(line, sw);
    uint16_t lvl = (uint16_t)(vbl ? BLANKING_LEVEL : BLACK_LEVEL);
    int n = (v->line_width - sw) >> 1; /* fill32 writes pairs */
    for (int i = 0; i < 2 * n; i++)
        line[sw + i] = lvl;
    if (!vbl)
        put_burst(v, line, line_counter);
}

static void put_pal_sync(const video_t* v, uint16_t* line, int i) /* video.cpp:918-934 */
{
    static const uint8_t sync_type[8] = {0, 0, 0, 3, 3, 2, 0, 0};
    int t = sync_type[i - 304];
    int half = v->line_width / 2;
    for (int h = 0; h < 2; h++) {
        int sw = (t & (h ? 1 : 2)) ? v->hsync_long : v->hsync_short;
        uint16_t* l = line + h * half;
        int k = 0;
        for (; k < sw; k++)
            l[k] = (uint16_t)SYNC_LEVEL;
        for (; k < half; k++)
            l[k] = (uint16_t)BLANKING_LEVEL;
    }
}

/* composite(), video.cpp:845-887: overlay text and progress bar on line `line` of the overlay */
static void put_overlay(const video_t* v, uint16_t* dst, int line, const uint8_t* overlay, int blend, int progress)
{
    if (!blend)
        return;
    if (v->pal)
        dst += 80;
    dst += 16; /* d32 += 8 */
    int scale = 255 / 4;
    if (blend != -1 && blend < 32)
        scale = (scale * blend) >> 5;
    for (int n = 0; n < 80; n++) {
        uint32_t p = BLACK_LEVEL + (overlay ? overlay[line * 80 + n] : 0) * scale;
        st32(dst, (p << 16) | p);
        dst += 2;
    }
    if (line < 3 || line > 8)
        return;
    dst += 16;
    uint32_t c0 = BLACK_LEVEL + (scale << 8), c1 = BLACK_LEVEL + (scale << 7);
    for (int i = 0; i < 352 - 80 - 32; i += 2) {
        uint32_t c = i < progress ? c0 : c1;
        st32(dst, (c << 16) | c);
        st32(dst + 2, (c << 16) | c);
        dst += 4;
    }
}

long efxo_video_field_ex(const uint8_t* frames2, int ntsc, int frame_counter0, int nfields, int front,
                         const int16_t* hscroll, const uint8_t* overlay, int blend, int progress, uint16_t* out)
{
    if (!frames2 || !out || nfields < 0 || front < 0 || front > 1)
        return -1;
    video_t v;
    video_setup(&v, ntsc);
    frame_t fr[2];
    for (int k = 0; k < 2; k++)
        for (int s = 0; s < STRIPS; s++)
            fr[k].strip[s] = (uint8_t*)frames2 + (size_t)k * EFXO_FRAME_BYTES + (size_t)s * STRIP_BYTES;
    /* two DMA line buffers ping-pong (video.cpp:171-186); active lines only rewrite sync,
     * burst and the picture window, the rest is what the last blanking line left there */
    uint16_t* bufs[2];
    bufs[0] = (uint16_t*)calloc((size_t)v.line_width, 2);
    bufs[1] = (uint16_t*)calloc((size_t)v.line_width, 2);
    int active_top = 32 + (v.pal ? 32 : 0);
    int active_bottom = active_top + 192;
    int vsync_start = v.line_count - (v.pal ? 8 : 3);
    int frame_counter = frame_counter0;
    for (int fld = 0; fld < nfields; fld++) {
        for (int i = 0; i < v.line_count; i++) { /* video_isr, video.cpp:1122-1198 */
            uint16_t* buf = bufs[i & 1];
            int line_counter = i + 1;
            if (i >= active_top && i < active_bottom) {
                put_sync(buf, v.hsync);
                put_burst(&v, buf, line_counter);
                uint16_t* dst = buf + v.active_start + 16;
                int f = front, h = hscroll ? hscroll[fld] : 0; /* video.cpp:1146-1154 */
                if (h < 0) {
                    h += 352;
                    f ^= 1;
                }
                blit_line(&v, &fr[f], dst, i - active_top, frame_counter, h, 352 - h);
                if (h)
                    blit_line(&v, &fr[f ^ 1], dst + (352 - h) * 2, i - active_top, frame_counter, 0, h);
            } else if (i >= vsync_start) {
                if (v.pal)
                    put_pal_sync(&v, buf, i);
                else
                    put_blanking(&v, buf, 1, line_counter);
            } else {
                put_blanking(&v, buf, 0, line_counter);
                int ptop = active_bottom + 2; /* video.cpp:1181-1187 */
                if (i >= ptop && i < ptop + 16)
                    put_overlay(&v, buf + v.active_start + 16, i - ptop, overlay, blend, progress);
            }
            memcpy(out + ((size_t)fld * v.line_count + i) * v.line_width, buf, (size_t)v.line_width * 2);
        }
        frame_counter++;
        if (blend > 0) /* video.cpp:1192-1193 */
            --blend;
    }
    free(bufs[0]);
    free(bufs[1]);
    return (long)v.line_count * v.line_width;
}

long efxo_video_field(const uint8_t* frames2, int ntsc, int frame_counter0, int nfields, uint16_t* out)
{
    return efxo_video_field_ex(frames2, ntsc, frame_counter0, nfields, 0, NULL, NULL, 0, 0, out);
}

/* ------------------------------------------------------------------------------------ */
/* PDM (espflix.ino:73-145)                                                             */

void efxo_pdm_second_order(int32_t state[3], uint16_t* dst, const int16_t* src, int len)
{
    const int32_t a1 = (int32_t)(0x7FFF * 1.18940);
    const int32_t a2 = (int32_t)(0x7FFF * 2.12340);
    uint32_t i0 = (uint32_t)state[0], i1 = (uint32_t)state[1], i2 = (uint32_t)state[2]; /* wrap like int32 */
    uint32_t bits = 0;
    int32_t s = 0;
    for (int half = 0; half < 2 * len; half++) {
        if (!(half & 1))
            s = *src++ * 2;
        i0 = (uint32_t)(((int32_t)(i0 + (uint32_t)s)) >> 1); /* one-pole interpolator */
        for (int n = 0; n < 16; n++) {
            bits <<= 1;
            uint32_t fb = (uint32_t)(((int32_t)i2) >> 7);
            if ((int32_t)i2 >= 0) {
                i1 += i0 - (uint32_t)a1 - fb;
                i2 += i1 - (uint32_t)a2;
                bits |= 1;
            } else {
                i1 += i0 + (uint32_t)a1 - fb;
                i2 += i1 + (uint32_t)a2;
            }
        }
        *dst++ = (uint16_t)bits;
    }
    state[0] = (int32_t)i0;
    state[1] = (int32_t)i1;
    state[2] = (int32_t)i2;
}

void efxo_write_pcm_16(int32_t state[3], int* beep, const int16_t* s, int n, uint16_t out256[256])
{
    /* _sin32 (espflix.ino:109-114) is -32767 sin(2 pi i / 32) truncated toward zero */
    int16_t sin32[32];
    for (int i = 0; i < 32; i++)
        sin32[i] = (int16_t)(-32767.0 * sin(2 * M_PI * i / 32));
    if (beep && *beep) {
        int16_t b[128];
        for (int i = 0; i < 128; i++)
            b[i] = (int16_t)(sin32[i & 31] >> 2);
        (*beep)--;
        efxo_pdm_second_order(state, out256, b, 128);
    } else if (s)
        efxo_pdm_second_order(state, out256, s, n);
    else
        for (int i = 0; i < 256; i++)
            out256[i] = 0xAAAA;
}

/* ------------------------------------------------------------------------------------ */
/* SBC audio (sbc_decoder.cpp)                                                          */

static const uint8_t sbc_block_mode[4] = {4, 8, 12, 16}; /* sbc_decoder.cpp:22 */
static const int8_t sbc_offset4[4][4] = {{-1, 0, 0, 0}, {-2, 0, 0, 1}, {-2, 0, 0, 1}, {-2, 0, 0, 1}};
static const int8_t sbc_offset8[4][8] = {{-2, 0, 0, 0, 0, 0, 0, 1},
                                         {-3, 0, 0, 0, 0, 0, 1, 2},
                                         {-4, 0, 0, 0, 0, 0, 1, 2},
                                         {-4, 0, 0, 0, 0, 0, 1, 2}}; /* A2DP 12.8 */

/* The reference tabulates the synthesis matrix and the prototype window as fixed-point words
 * (sbc_decoder.cpp:41-71).  They are the A2DP (Appendix B) definitions
 *   syn[i][k]      = floor(65536 cos((i + 4)(2k + 1) pi / 16))               16 x 8
 *   m[i][j]        = floor(-8 * 32768 * Proto_8_80[8 j + i])                  8 x 10
 * with Proto_8_80 the standard's 80-tap prototype filter (symmetric about tap 40 except that
 * taps 48 and 64 are the negatives of taps 32 and 16).  Generated here from those
 * definitions; test_oracle_vs_ref pins every word against the reference's own tables. */
static const double sbc_proto_8_80_half[41] = {
    0.00000000E+00, 1.56575398E-04, 3.43256425E-04, 5.54620202E-04, 8.23919506E-04, 1.13992507E-03,
    1.47640169E-03, 1.78371725E-03, 2.01182542E-03, 2.10371989E-03, 1.99454554E-03, 1.61656283E-03,
    9.02154502E-04, -1.78805361E-04, -1.64973098E-03, -3.49717454E-03, 5.65949473E-03, 8.02941163E-03,
    1.04584443E-02, 1.27472335E-02, 1.46525263E-02, 1.59045603E-02, 1.62208471E-02, 1.53184106E-02,
    1.29371806E-02, 8.85757540E-03, 2.92408442E-03, -4.91578024E-03, -1.46404076E-02, -2.61098752E-02,
    -3.90751381E-02, -5.31873032E-02, 6.79989431E-02, 8.29847578E-02, 9.75753918E-02, 1.11196689E-01,
    1.23264548E-01, 1.33264415E-01, 1.40753505E-01, 1.45389847E-01, 1.46955068E-01};
static int32_t sbc_syn_8[128], sbc_proto_8[80];
static int sbc_tables_ready;

static void sbc_tables_init(void)
{
    if (sbc_tables_ready)
        return;
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 8; k++) {
            double x = cos((i + 4) * (2 * k + 1) * M_PI / 16) * 65536.0;
            sbc_syn_8[i * 8 + k] = fabs(x) < 1e-6 ? 0 : (int32_t)floor(x);
        }
    for (int n = 0; n < 80; n++) {
        int h = n <= 40 ? n : 80 - n;
        double p = sbc_proto_8_80_half[h];
        if (n > 40 && (h == 16 || h == 32))
            p = -p;
        int32_t d = p == 0 ? 0 : (int32_t)floor(-8.0 * 32768.0 * p);
        sbc_proto_8[(n & 7) * 10 + (n >> 3)] = d;
    }
    sbc_tables_ready = 1;
}

void efxo_sbc_tables(int32_t syn128[128], int32_t proto80[80])
{
    sbc_tables_init();
    memcpy(syn128, sbc_syn_8, sizeof(sbc_syn_8));
    memcpy(proto80, sbc_proto_8, sizeof(sbc_proto_8));
}

static void sbc_synthesize8(int32_t* v, uint8_t* offset, const int32_t* src, int16_t* dst) /* sbc_decoder.cpp:74-139 */
{
    const int32_t* syn = sbc_syn_8;
    for (int i = 0; i < 16; i++) { /* matrixing: 128 MACs, 32-bit wrapping accumulate */
        if (!offset[i]) {
            for (int j = 0; j < 9; j++)
                v[j + 160] = v[j];
            offset[i] = 160;
        }
        int k = --offset[i];
        uint32_t s = 0;
        for (int j = 0; j < 8; j++)
            s += (uint32_t)syn[j] * (uint32_t)src[j];
        syn += 8;
        v[k] = (int32_t)s >> 15;
    }
    const int32_t* m = sbc_proto_8;
    for (int i = 0; i < 8; i++) { /* windowing: 80 MACs */
        const int32_t* p0 = v + offset[i];
        const int32_t* p1 = v + offset[(i + 8) & 0xF] + 1;
        uint32_t s = 0;
        for (int j = 0; j < 10; j += 2) {
            s += (uint32_t)p0[j] * (uint32_t)m[j];
            s += (uint32_t)p1[j] * (uint32_t)m[j + 1];
        }
        m += 10;
        int32_t r = (int32_t)s >> 15;
        if (r < -0x7FFF)
            r = -0x7FFF;
        else if (r > 0x7FFF)
            r = 0x7FFF;
        dst[i] = (int16_t)r;
    }
}

static void sbc_bit_allocation(const efxo_sbc* sbc, uint8_t (*scale_factor)[8], int (*bits)[8]) /* sbc_decoder.cpp:142-240 */
{
    int sf = sbc->frequency, bp = sbc->bitpool, nsb = sbc->subbands;
    int bitneed[2][8];
    for (int ch = 0; ch < sbc->channels; ch++) {
        int max_bitneed = 0;
        for (int sb = 0; sb < nsb; sb++) {
            int s = scale_factor[ch][sb];
            if (sbc->allocation) /* SNR */
                bitneed[ch][sb] = s;
            else if (s == 0) /* loudness */
                bitneed[ch][sb] = -5;
            else {
                int loudness = s - (nsb == 4 ? sbc_offset4[sf][sb] : sbc_offset8[sf][sb]);
                if (loudness > 0)
                    loudness /= 2;
                bitneed[ch][sb] = loudness;
            }
            if (bitneed[ch][sb] > max_bitneed)
                max_bitneed = bitneed[ch][sb];
        }
        int bitcount = 0, slicecount = 0, bitslice = max_bitneed + 1;
        do {
            bitslice--;
            bitcount += slicecount;
            slicecount = 0;
            for (int sb = 0; sb < nsb; sb++) {
                if (bitneed[ch][sb] > bitslice + 1 && bitneed[ch][sb] < bitslice + 16)
                    slicecount++;
                else if (bitneed[ch][sb] == bitslice + 1)
                    slicecount += 2;
            }
        } while (bitcount + slicecount < bp);
        if (bitcount + slicecount == bp) {
            bitcount += slicecount;
            bitslice--;
        }
        for (int sb = 0; sb < nsb; sb++) {
            if (bitneed[ch][sb] < bitslice + 2)
                bits[ch][sb] = 0;
            else {
                bits[ch][sb] = bitneed[ch][sb] - bitslice;
                if (bits[ch][sb] > 16)
                    bits[ch][sb] = 16;
            }
        }
        for (int sb = 0; bitcount < bp && sb < nsb; sb++) {
            if (bits[ch][sb] >= 2 && bits[ch][sb] < 16) {
                bits[ch][sb]++;
                bitcount++;
            } else if (bitneed[ch][sb] == bitslice + 1 && bp > bitcount + 1) {
                bits[ch][sb] = 2;
                bitcount += 2;
            }
        }
        for (int sb = 0; bitcount < bp && sb < nsb; sb++) {
            if (bits[ch][sb] < 16) {
                bits[ch][sb]++;
                bitcount++;
            }
        }
    }
}

static int sbc_get_samples(efxo_sbc* sbc, const uint8_t* data, int len) /* sbc_decoder.cpp:276-344 */
{
    int bits[2][8];
    uint8_t scale_factor[2][8];
    if (len < 4 || data[0] != 0x9C)
        return -1;
    sbc->frequency = (data[1] >> 6) & 3;
    sbc->blocks = sbc_block_mode[(data[1] >> 4) & 3];
    sbc->mode = (data[1] >> 2) & 3;
    sbc->channels = !sbc->mode ? 1 : 2;
    sbc->allocation = (data[1] >> 1) & 1;
    sbc->subbands = (data[1] & 1) ? 8 : 4;
    sbc->bitpool = data[2];
    if (sbc->mode == 3 || sbc->subbands == 4) /* CRC ignored */
        return -1;
    /* A bitpool above 16 bits x 8 subbands can never be met: the reference's allocation loop
     * (sbc_decoder.cpp:186-198) then never terminates.  Defined here: the frame is rejected like
     * a joint-stereo one. */
    if (sbc->bitpool > 128)
        return -1;
    const uint8_t* sf = data + 4;
    for (int ch = 0; ch < sbc->channels; ch++)
        for (int sb = 0; sb < sbc->subbands; sb += 2) {
            uint8_t a = *sf++;
            scale_factor[ch][sb] = a >> 4;
            scale_factor[ch][sb + 1] = a & 0xF;
        }
    sbc_bit_allocation(sbc, scale_factor, bits);
    int b_count = 0;
    uint32_t b_bits = 0;
    const uint8_t* b_data = data + 4 + (sbc->channels * sbc->subbands >> 1);
    for (int blk = 0; blk < sbc->blocks; blk++)
        for (int ch = 0; ch < sbc->channels; ch++)
            for (int sb = 0; sb < sbc->subbands; sb++) {
                int32_t sample = 0;
                int level = bits[ch][sb];
                if (level) {
                    while (b_count < level) {
                        b_bits = (b_bits << 8) | *b_data++;
                        b_count += 8;
                    }
                    b_count -= level;
                    sample = (int32_t)((b_bits >> b_count) & ((1u << level) - 1));
                    int scale = scale_factor[ch][sb];
                    /* IQUANT (sbc_decoder.cpp:263-270): 32-bit wrapping shift, C division */
                    sample = (sample << 1) | 1;
                    sample = (int32_t)((uint32_t)sample << scale) / ((1 << level) - 1);
                    sample -= 1 << scale;
                }
                sbc->sb_sample[blk][ch][sb] = sample;
            }
    return (int)(b_data - data);
}

void efxo_sbc_init(efxo_sbc* s) { memset(s, 0, sizeof(*s)); } /* sbc_decoder.cpp:375-378 */

int efxo_sbc_decode(efxo_sbc* sbc, const uint8_t* src, int src_len, int16_t* dst, int* decoded) /* sbc_decoder.cpp:346-373 */
{
    sbc_tables_init();
    if (!sbc->inited) {
        sbc->inited = 1;
        for (int ch = 0; ch < 2; ch++)
            for (int i = 0; i < 16; i++)
                sbc->v_offset[ch][i] = (uint8_t)((i + 1) * 10);
    }
    int framelen = sbc_get_samples(sbc, src, src_len);
    if (sbc->subbands == 4)
        return -1;
    /* a frame get_samples() rejected is still synthesised -- from the subband samples and the
     * geometry the state holds (the previous frame's, or zeros) */
    int16_t* pcm = dst;
    for (int ch = 0; ch < sbc->channels; ch++)
        for (int blk = 0; blk < sbc->blocks; blk++) {
            sbc_synthesize8(sbc->v[ch], sbc->v_offset[ch], sbc->sb_sample[blk][ch], pcm);
            pcm += 8;
        }
    if (decoded)
        *decoded = sbc->blocks * sbc->subbands * sbc->channels * 2;
    return framelen;
}

size_t efxo_ts_audio_es(const uint8_t* ts, size_t len, uint8_t* out, size_t cap) /* player.cpp:381-433,459-493 */
{
    size_t n = 0;
    int64_t audio_pts = -1;
    for (size_t pos = 0; pos + 188 <= len; pos += 188) {
        const uint8_t* p = ts + pos;
        if (p[0] != 0x47)
            continue;
        int pid = ((p[1] << 8) + p[2]) & 0x1fff;
        const uint8_t* pay = p + 4;
        if (p[3] & 0x20)
            pay = p + 5 + p[4];
        if (!(p[3] & 0x10))
            continue;
        const uint8_t* pend = p + 188;
        int64_t pts = -1;
        int start = p[1] & 0x40;
        if (start) {
            if (pay + 9 > pend) /* defined as for video: a PES header that does not fit drops the packet */
                continue;
            const uint8_t* q = pay + 6;
            int flags = (q[0] << 8) | q[1];
            pay = q + 3 + q[2];
            q += 3;
            if ((flags & 0x0080) && q + 5 <= pend)
                pts = parse_pts(q, flags);
        }
        if (pid != 0x101 && pid != 0x102)
            continue;
        if (start)
            audio_pts = pts;
        if (audio_pts != -1 && pay < pend)
            for (const uint8_t* b = pay; b < pend; b++, n++)
                if (n < cap)
                    out[n] = *b;
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* trick-play index (indexer/indexer.cpp, espflix.cpp:573-629)                          */

long efxo_ts_sequences(const uint8_t* ts, size_t len, int64_t* first_pts, int64_t* last_pts, int64_t* seq_pts,
                       uint32_t* seq_pos, long cap) /* indexer.cpp:86-176 */
{
    int64_t origin = -1, video_pts = -1;
    long n = 0;
    uint32_t packet = 0;
    for (size_t pos = 0; pos + 188 <= len; pos += 188, packet++) {
        const uint8_t* d = ts + pos;
        int pid = ((d[1] << 8) + d[2]) & 0x1fff;
        const uint8_t* data = d + 4;
        if (d[3] & 0x20)
            data = d + 5 + d[4];
        if (!(d[3] & 0x10) || !(d[1] & 0x40))
            continue;
        /* parse(), indexer.cpp:54-74: pts = 0 without a PTS flag, -1 on a marker mismatch; the
         * "marker" is the fourth payload byte (the start code value).  The indexer reads these
         * bytes wherever they fall; a header that leaves the packet is skipped here. */
        const uint8_t* end = d + 188;
        if (data + 9 > end)
            continue;
        const uint8_t* q = data + 6;
        int flags = (q[0] << 8) | q[1];
        const uint8_t* payload = q + 3 + q[2];
        q += 3;
        int64_t pts = 0;
        if (flags & 0x0080)
            pts = q + 5 <= end ? parse_pts(q, flags) : -1;
        int m = payload + 3 < end ? payload[3] : -1;
        if (pid != 0x100)
            continue;
        if (m == 0xB3) {
            if (origin == -1)
                origin = pts;
            if (n < cap) {
                seq_pts[n] = pts;
                seq_pos[n] = packet;
            }
            n++;
        }
        video_pts = pts;
    }
    *first_pts = origin;
    *last_pts = video_pts;
    return n;
}

static uint32_t idx_pts2pos(int64_t pts, const int64_t* seq_pts, const uint32_t* seq_pos, long n) /* indexer.cpp:182-195 */
{
    long mini = 0;
    int mine = 0x7FFFFFF;
    for (long i = 0; i < n; i++) {
        int64_t dd = seq_pts[i] - pts;
        int e = (int)(uint32_t)(uint64_t)(dd < 0 ? -dd : dd); /* (int)abs(...) of an int64 */
        if (e < mine) {
            mine = e;
            mini = i;
        }
    }
    return seq_pos[mini];
}

size_t efxo_make_idx(const uint8_t* const ts[3], const size_t len[3], uint8_t* out, size_t cap) /* indexer.cpp:197-237 */
{
    efxo_idx_rec rec[3];
    uint32_t* samples[3] = {0, 0, 0};
    const uint32_t speed[3] = {1, 15, 15};
    size_t total = 8 + 3 * 32;
    int bad = 0;
    for (int k = 0; k < 3; k++) {
        long cap_seq = (long)(len[k] / 188) + 1;
        int64_t* sp = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap_seq);
        uint32_t* so = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap_seq);
        int64_t first, last;
        long n = efxo_ts_sequences(ts[k], len[k], &first, &last, sp, so, cap_seq);
        memset(&rec[k], 0, sizeof(rec[k]));
        if (n <= 0)
            bad = 1;
        else {
            const uint32_t bin = 90000 / 12;
            int64_t end = last - first;
            size_t count = end < 0 ? 0 : (size_t)(end / bin) + 1;
            samples[k] = (uint32_t*)malloc(4 * (count ? count : 1));
            size_t c = 0;
            for (int64_t pts = 0; pts <= end; pts += bin) /* pts2seq, indexer.cpp:197-217 */
                samples[k][c++] = idx_pts2pos(pts + first, sp, so, n);
            rec[k].first_pts = first;
            rec[k].last_pts = last;
            rec[k].sample_count = (uint32_t)c;
            rec[k].bin_size = bin;
            rec[k].trick_speed = speed[k];
            total += 4 * c;
        }
        free(sp);
        free(so);
    }
    if (!bad && total <= cap) {
        uint32_t sig = 'I' | ('D' << 8) | ('X' << 16), three = 3;
        uint8_t* p = out;
        memcpy(p, &sig, 4);
        memcpy(p + 4, &three, 4);
        p += 8;
        for (int k = 0; k < 3; k++) {
            /* idx_rec as the compiler lays it out: 28 bytes of fields + 4 of tail padding, which
             * fwrite() of the struct writes out as whatever the stack held -- compared masked */
            memcpy(p, &rec[k], 32);
            p += 32;
        }
        for (int k = 0; k < 3; k++) {
            memcpy(p, samples[k], 4 * (size_t)rec[k].sample_count);
            p += 4 * (size_t)rec[k].sample_count;
        }
    }
    for (int k = 0; k < 3; k++)
        free(samples[k]);
    return bad ? 0 : total;
}

static void idx_load(const uint8_t* hdr, efxo_idx_rec r[3])
{
    memcpy(r, hdr + 8, 96);
}

static int64_t idx_map_pts(int64_t pts, const efxo_idx_rec* r, const efxo_idx_rec* video) /* espflix.cpp:589-594 */
{
    pts -= r->first_pts;
    pts *= video->last_pts - video->first_pts;
    return pts / (r->last_pts - r->first_pts);
}

int64_t efxo_idx_pts2pts(const uint8_t* hdr, int64_t pts, int speed) /* espflix.cpp:597-604 */
{
    efxo_idx_rec r[3];
    idx_load(hdr, r);
    if (speed == 1)
        return r[0].first_pts + idx_map_pts(pts, &r[1], &r[0]);
    if (speed == -1)
        return r[0].last_pts - idx_map_pts(pts, &r[2], &r[0]);
    return pts;
}

uint32_t efxo_idx_pts2offset(const uint8_t* hdr, int64_t pts, int speed) /* espflix.cpp:607-627 */
{
    efxo_idx_rec r[3];
    idx_load(hdr, r);
    const efxo_idx_rec *video = &r[0], *fwd = &r[1], *rwd = &r[2];
    uint32_t offset;
    if (pts > video->last_pts)
        pts = video->last_pts;
    if (pts < video->first_pts)
        pts = video->first_pts;
    if (speed == 1) {
        offset = (uint32_t)(pts - video->first_pts) / fwd->trick_speed / fwd->bin_size;
        if (fwd->sample_count - 1 < offset)
            offset = fwd->sample_count - 1;
        offset += video->sample_count;
    } else if (speed == -1) {
        offset = (uint32_t)((video->last_pts - pts) - video->first_pts) / rwd->trick_speed / rwd->bin_size;
        if (rwd->sample_count - 1 < offset)
            offset = rwd->sample_count - 1;
        offset += video->sample_count + fwd->sample_count;
    } else {
        offset = (uint32_t)((pts - video->first_pts) / video->bin_size);
        if (video->sample_count - 1 < offset)
            offset = video->sample_count - 1;
    }
    return offset * 4 + 104;
}
