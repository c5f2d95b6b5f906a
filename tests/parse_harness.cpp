// tests/parse_harness.cpp -- TEST TOOL: runs the lane-level slice parser of k_parse (espflix_amd/csrc/parse_tm.h, compiled
// here for the host) over every slice of an elementary stream and checks each macroblock record and coefficient entry
// against the parse trace of the test oracle (oracle/efx_oracle.c, efxo_set_trace) decoding the same stream.  CPU only: the
// token machine's tables and both of its passes are verified without a GPU; the -m gpu tests then check the kernel built
// from the same header against the frames of the reference.
//
//   parse_harness <file> [ts]     exit 0 and "OK slices=.. macroblocks=.. entries=.. trips=.. rejected=.. unseen=.. phantom=.." or a mismatch report
//   parse_harness --selftest      hand-built slices for what no decodable stream reaches: a slice whose words outgrow its region
//                                 (pass 2 must not touch the next slice's slots), address escapes without end, long runs of
//                                 macroblock_stuffing
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "efx_oracle.h"
}
#include "parse_tm.h"

using namespace efx;

namespace {

struct Ev {
    int kind, a, b, c, e;
};
std::vector<Ev> g_ev;
void on_trace(void*, int kind, int a, int b, int c, int e) { g_ev.push_back({kind, a, b, c, e}); }

struct HostBits {
    const uint8_t* base;
    uint32_t pos;
    uint32_t window() const
    {
        const uint8_t* q = base + (pos >> 3);
        uint64_t w = 0;
        for (int i = 0; i < 8; i++)
            w = (w << 8) | q[i];
        return (uint32_t)((w << (pos & 7)) >> 32);
    }
};

struct ExpMb {
    int addr;
    uint32_t flags;
    int mvx, mvy;
    uint32_t cnt[6];
    std::vector<uint32_t> entries;
};

struct BitWriter {
    std::vector<uint8_t> bytes;
    int fill = 0;
    void put(uint32_t v, int n)
    {
        for (int i = n - 1; i >= 0; i--) {
            if (!fill)
                bytes.push_back(0);
            bytes.back() |= (uint8_t)(((v >> i) & 1) << (7 - fill));
            fill = (fill + 1) & 7;
        }
    }
};

// One slice (the bits after quantiser_scale / extra_bit_slice), run exactly as the harness runs a slice of a stream.
struct Solo {
    std::vector<uint32_t> coefs;
    std::vector<MbRec> recs;
    uint32_t status = 0, n_mbs = 0, why = 0;
    TmLane L;
};
Solo run_solo(const TmTables* tab, const BitWriter& bw, bool intra_picture, uint32_t region_slots, int code = 1, int mb_limit = kMbCount)
{
    Solo o;
    std::vector<uint8_t> es = bw.bytes;
    es.resize(es.size() + 64, 0);
    const uint32_t tok_base = 64;
    o.coefs.assign(tok_base + region_slots + 4096, 0xDEADBEEFu);
    o.recs.assign(kMbCount, MbRec{});
    std::vector<TmU4> raw(kMbCount + 1);
    HostBits br{es.data(), 0};
    TmSlice sp;
    sp.coef_last = tok_base + region_slots - 1;
    sp.type_bit = intra_picture ? kTmTypeIBit : 0u;
    sp.r_size = 0;
    sp.max_mbs = (uint32_t)(mb_limit - (code - 1) * kMbW);
    TmFix fx{};
    fx.code = code;
    fx.mb_limit = mb_limit;
    fx.qscale = 4;
    fx.epoch = 9;
    fx.coef_last = sp.coef_last;
    tm_begin(o.L, tok_base, true);
    auto store_raw = [&](uint32_t k, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { raw[k] = TmU4{a, b, c, d}; };
    long trips = 0;
    while (tm_alive(o.L.st) && trips < 10000000) {
        if ((trips & 3) == 0)
            tm_guard(o.L);
        const uint32_t win = br.pos / 8 + 8 < es.size() ? br.window() : 0u;
        const TmE e = tab->e[(o.L.st >> 19) + (win >> (o.L.st & 31))];
        if (tm_trip(o.L, win, e, sp, [&](uint32_t bits, uint32_t) { br.pos += bits; }, TmDirectSink{o.coefs.data(), sp.coef_last}, store_raw))
            br.pos -= tm_overflow(o.L, e, sp, TmDirectSink{o.coefs.data(), sp.coef_last}, store_raw);
        trips++;
    }
    tm_end(o.L, sp, store_raw);
    uint32_t nc = 0;
    o.status = tm_finish(o.L, fx, tok_base, [&](uint32_t k) { return raw[k]; }, o.coefs.data(), reinterpret_cast<TmU4*>(o.recs.data()), &o.n_mbs, &nc);
    o.why = tm_why(o.L.st);
    return o;
}

int selftest()
{
    TmTables* tab = new TmTables;
    build_tm_tables(tab);
    int bad = 0;
    auto expect = [&](bool ok, const char* what) {
        if (!ok) {
            fprintf(stderr, "selftest: %s\n", what);
            bad++;
        }
    };
    // ---- a slice whose words outgrow its region ---------------------------------------------------------------------------
    // one intra macroblock of 6 blocks x (DC + three coefficients) = 24 stream words in a region of 8 slots: pass 1 drops the
    // words beyond the region, pass 2 must not write DC values (or shift words) there -- they are the next slice's
    {
        BitWriter bw;
        bw.put(1, 1);  // macroblock_address_increment 1
        bw.put(1, 1);  // macroblock_type I: intra
        for (int b = 0; b < 6; b++) {
            if (b < 4)
                bw.put(0x1, 3);  // dct_dc_size_luminance 1 ("00") + differential bit 1
            else
                bw.put(0x0, 2);  // dct_dc_size_chrominance 0 ("00")
            for (int k = 0; k < 3; k++)
                bw.put(0x6, 3);  // "11" + sign 0: run 0, level 1
            bw.put(0x2, 2);      // end_of_block
        }
        Solo o = run_solo(tab, bw, true, 8);
        expect((o.status & EFX_STREAM_BAD_VLC) != 0, "ran past its region: EFX_STREAM_BAD_VLC expected");
        bool clean = true;
        for (size_t i = 64 + 8; i < o.coefs.size(); i++)
            clean = clean && o.coefs[i] == 0xDEADBEEFu;
        expect(clean, "ran past its region: a slot beyond the region was written");
        expect(o.recs[0].epoch == 0, "ran past its region: the macroblock must not be kept");
        // the same macroblock in a region that holds it: kept, DC values in place
        Solo f = run_solo(tab, bw, true, 28);
        expect(f.status == 0 && f.n_mbs == 1 && f.recs[0].epoch == 9 && f.recs[0].cnt[0] == 4 && f.recs[0].cnt[5] == 4,
               "one intra macroblock in a region that holds it");
        expect(f.coefs[64] == (uint32_t)(128 + 1) << 6 && f.coefs[64 + 16] == (uint32_t)128 << 6, "DC values of the intra macroblock");
        // an invalid code inside a block, met beyond the region: the blocks before it must not be written either
        BitWriter bx;
        bx.put(1, 1);
        bx.put(1, 1);
        for (int b = 0; b < 3; b++) {
            bx.put(0x1, 3);
            for (int k = 0; k < 3; k++)
                bx.put(0x6, 3);
            bx.put(0x2, 2);
        }
        bx.put(0x1, 3);
        bx.put(0x6, 3);
        bx.put(0x0, 12);  // twelve zero bits: no such coefficient code ... (the tail of zeros follows)
        Solo x = run_solo(tab, bx, true, 8);
        clean = true;
        for (size_t i = 64 + 8; i < x.coefs.size(); i++)
            clean = clean && x.coefs[i] == 0xDEADBEEFu;
        expect((x.status & EFX_STREAM_BAD_VLC) != 0 && clean, "bad code beyond the region: flagged, nothing written past the region");
    }
    // ---- macroblock_stuffing, any number of codes (player.cpp:1268-1270 simply loops) -------------------------------------------
    for (int n_stuff : {1, 7, 8, 15, 16, 17, 40, 63, 64, 65, 127, 128, 129, 255, 256, 300, 1000}) {
        BitWriter bw;
        bw.put(1, 1);    // increment 1
        bw.put(1, 3);    // macroblock_type P "001": motion forward, no pattern
        bw.put(1, 1);    // motion_horizontal_forward_code 0
        bw.put(1, 1);    // motion_vertical_forward_code 0
        for (int i = 0; i < n_stuff; i++)
            bw.put(0xF, 11);  // macroblock_stuffing 0000 0001 111
        bw.put(1, 1);
        bw.put(1, 3);
        bw.put(1, 1);
        bw.put(1, 1);
        Solo o = run_solo(tab, bw, false, 4096);
        char what[96];
        snprintf(what, sizeof what, "%d stuffing codes before a macroblock: two records, zero vectors, status 0", n_stuff);
        expect(o.status == 0 && o.n_mbs == 2 && o.recs[1].epoch == 9 && o.recs[1].mvx == 0 && o.recs[1].mvy == 0 && o.why == kDeadEnd, what);
        // ... and stuffing followed by the end of the slice is not where a slice may end (slice_done() is asked between
        // macroblocks only): flagged, whatever the count
        BitWriter be;
        be.put(1, 1);
        be.put(1, 3);
        be.put(1, 1);
        be.put(1, 1);
        for (int i = 0; i < n_stuff; i++)
            be.put(0xF, 11);
        Solo z = run_solo(tab, be, false, 4096);
        snprintf(what, sizeof what, "%d stuffing codes, then the end of the slice: EFX_STREAM_BAD_VLC", n_stuff);
        expect((z.status & EFX_STREAM_BAD_VLC) != 0, what);
    }
    // ---- address escapes without end: the increment must not wrap into a small one --------------------------------------------
    for (int n_esc : {7, 8, 2000, 40000, 70000}) {
        BitWriter bw;
        bw.put(1, 1);
        bw.put(1, 3);
        bw.put(1, 1);
        bw.put(1, 1);
        for (int i = 0; i < n_esc; i++)
            bw.put(0x8, 11);  // macroblock_escape 0000 0001 000
        bw.put(1, 1);
        bw.put(1, 3);
        bw.put(1, 1);
        bw.put(1, 1);
        Solo o = run_solo(tab, bw, false, 1u << 20);
        char what[96];
        snprintf(what, sizeof what, "%d address escapes: one record kept, EFX_STREAM_MB_OVERRUN", n_esc);
        // (7 escapes + 1 = 232 more macroblocks: inside the picture; from 8 on the address lies beyond it)
        if (n_esc == 7)
            expect(o.status == 0 && o.n_mbs == 233 && o.recs[232].epoch == 9, "7 address escapes: macroblock 232");
        else
            expect((o.status & EFX_STREAM_MB_OVERRUN) != 0 && o.recs[0].epoch == 9, what);
    }
    delete tab;
    if (!bad)
        printf("SELFTEST OK\n");
    return bad ? 1 : 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc >= 2 && !strcmp(argv[1], "--selftest"))
        return selftest();
    if (argc < 2) {
        fprintf(stderr, "usage: parse_harness <file> [ts]\n");
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
        in.insert(in.end(), buf, buf + n);
    fclose(f);
    const bool ts = argc > 2 && !strcmp(argv[2], "ts");
    std::vector<uint8_t> es;
    if (ts) {
        es.resize(in.size());
        es.resize(efxo_ts_to_es(in.data(), in.size(), es.data(), es.size()));
    } else
        es = in;
    const size_t es_len = es.size();
    static const uint8_t tail[kEsTailBytes] = {0, 0, 0, 1, 0xB7, 0, 0, 1, 0xB7};
    es.insert(es.end(), tail, tail + kEsTailBytes);
    es.resize(es.size() + kEsGuardBytes, 0);

    efxo_set_trace(on_trace, nullptr);
    efxo_decode(in.data(), in.size(), ts ? 1 : 0, 1, nullptr, nullptr, nullptr, 0);
    efxo_set_trace(nullptr, nullptr);

    // start codes in stream order (byte aligned, up to the first sequence_end_code)
    struct Unit {
        uint32_t off;
        int code;
    };
    std::vector<Unit> units;
    for (size_t i = 0; i + 3 < es_len + kEsTailBytes; i++)
        if (es[i] == 0 && es[i + 1] == 0 && es[i + 2] == 1) {
            units.push_back({(uint32_t)i + 4, es[i + 3]});
            if (es[i + 3] == 0xB7)
                break;
            i += 3;
        }

    TmTables* tab = new TmTables;
    build_tm_tables(tab);
    std::vector<uint32_t> coefs(es.size() * kCoefsPerEsByte + 16);
    std::vector<MbRec> recs(kMbCount);
    std::vector<TmU4> raw(kMbCount + 1);

    // The oracle's slices, each with the stream position its marker hunt stopped at (EFXO_T_SLICE_AT).  The reference hunts
    // for markers bit by bit and acts on every fourth byte of what it swallows (player.cpp:1360-1363): on a damaged stream,
    // and inside user data / extension payloads, it can see slices that have no byte-aligned start code behind them
    // (`phantom`), and run through start codes without acting on them (`unseen`).  Only a slice both sides see is compared.
    struct OSlice {
        size_t ev;     // index of its EFXO_T_SLICE event
        uint64_t bit;  // first bit after the marker, ~0 when the reader had reached the end pad
    };
    std::vector<OSlice> oslices;
    for (size_t q = 0; q < g_ev.size(); q++)
        if (g_ev[q].kind == EFXO_T_SLICE) {
            const bool at = q > 0 && g_ev[q - 1].kind == EFXO_T_SLICE_AT;
            oslices.push_back({q, at ? (uint64_t)g_ev[q - 1].a * 8 + (uint64_t)g_ev[q - 1].b : ~0ull});
        }
    size_t os = 0;
    size_t ev = 0;
    long slices = 0, mbs = 0, entries = 0, rejected = 0, unseen = 0, phantom = 0, trips = 0;
    const uint32_t epoch = 7;
    for (size_t u = 0; u < units.size(); u++) {
        const int code = units[u].code;
        if (code < 0x01 || code > 0xAF)
            continue;
        const uint64_t here = (uint64_t)units[u].off * 8;
        while (os < oslices.size() && oslices[os].bit < here) {
            os++;
            phantom++;
        }
        if (os >= oslices.size() || oslices[os].bit != here) {
            if (getenv("EFX_HARNESS_VERBOSE"))
                fprintf(stderr, "slice unit at %u code %02x: not seen by the oracle\n", units[u].off, code);
            unseen++;
            continue;
        }
        const Ev se = g_ev[oslices[os].ev];
        ev = oslices[os].ev + 1;
        os++;
        if (se.b != code) {
            fprintf(stderr, "slice unit at %u code %02x: the oracle's slice there is %02x\n", units[u].off, code, se.b);
            return 1;
        }
        if (se.c < 0) {
            // forward_f_code 0 in a (phantom) P header: the reference's forward_r_size is -1 and it shifts by it (undefined);
            // k_index clamps to 0 -- nothing to compare
            unseen++;
            continue;
        }
        const bool decoded = (se.c >> 16) & 1;
        if (se.a < 0 || !decoded || code - 2 >= kMbH) {
            rejected++;
            continue;
        }
        // the slice stops at the first macroblock of the next slice of its picture (k_slice_emit's rule, for slices in order)
        int mb_limit = kMbCount;
        for (size_t v = u + 1; v < units.size(); v++) {
            if (units[v].code >= 0x01 && units[v].code <= 0xAF) {
                if (units[v].code > code)
                    mb_limit = (units[v].code - 1) * kMbW < kMbCount ? (units[v].code - 1) * kMbW : kMbCount;
                break;
            }
            if (units[v].code == 0x00 || units[v].code == 0xB7 || units[v].code == 0xB3 || units[v].code == 0xB8)
                break;
        }
        const uint32_t next = (u + 1 < units.size()) ? units[u + 1].off - 4 : (uint32_t)(es_len + kEsTailBytes);
        const uint32_t len = next - units[u].off;
        // expected macroblocks of this slice
        std::vector<ExpMb> exp;
        std::vector<uint32_t> blk_entries;
        size_t e2 = ev, e_cut = 0;
        std::vector<size_t> mb_event;
        for (; e2 < g_ev.size() && g_ev[e2].kind != EFXO_T_SLICE; e2++) {
            const Ev& x = g_ev[e2];
            if (x.kind == EFXO_T_MB) {
                mb_event.push_back(e2);
                ExpMb m{};
                m.addr = x.a;
                m.flags = (uint32_t)x.b;
                m.mvx = x.c;
                m.mvy = x.e;
                exp.push_back(m);
                blk_entries.clear();
            } else if (x.kind == EFXO_T_COEF) {
                blk_entries.push_back(((uint32_t)x.c << 6) | (uint32_t)x.b);
            } else if (x.kind == EFXO_T_BLOCK) {
                if (x.b == 0) {
                    exp.back().cnt[x.a] = (uint32_t)blk_entries.size();
                    exp.back().entries.insert(exp.back().entries.end(), blk_entries.begin(), blk_entries.end());
                }
                blk_entries.clear();
            }
        }
        // (a damaged slice that runs on: the oracle, one serial decoder, follows it into the rows of the next slice; a parse
        // lane stops where that slice starts -- SliceDesc::mb_limit -- so only the macroblocks before it are compared)
        e_cut = e2;
        for (size_t q = 0; q < exp.size(); q++)
            if (exp[q].addr >= mb_limit) {
                exp.resize(q);
                e_cut = mb_event[q];
                break;
            }
        for (auto& r : recs)
            memset(&r, 0, sizeof r);

        // ---- the parser, as k_parse runs it ----------------------------------------------------------------------------
        HostBits br{es.data(), units[u].off * 8};
        TmSlice sp;
        sp.coef_last = (units[u].off + len) * kCoefsPerEsByte - 1;
        sp.type_bit = (se.c & 15) == 1 ? kTmTypeIBit : 0u;
        sp.r_size = (se.c >> 8) & 7;
        sp.max_mbs = (uint32_t)(mb_limit - (code - 1) * kMbW);
        TmFix fx;
        fx.code = code;
        fx.mb_limit = mb_limit;
        fx.full_pel = (se.c >> 4) & 1;
        fx.r_size = sp.r_size;
        fx.rec_flags = se.e ? 0x80u : 0u;
        fx.epoch = epoch;
        fx.coef_last = sp.coef_last;
        {
            uint32_t w = br.window();
            fx.qscale = w >> 27;
            br.pos += 5;
            while (br.window() >> 31)  // extra_bit_slice, player.cpp:1261-1262
                br.pos += 9;
            br.pos += 1;
        }
        TmLane L;
        const uint32_t tok_base = units[u].off * kCoefsPerEsByte;
        tm_begin(L, tok_base, true);
        auto store_raw = [&](uint32_t k, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { raw[k] = TmU4{a, b, c, d}; };
        long guard = 0;
        const long trips0 = trips;
        while (tm_alive(L.st)) {
            if ((trips & 3) == 0)
                tm_guard(L);  // (k_parse: between two groups of trips)
            const uint32_t win = br.window();
            const TmE e = tab->e[(L.st >> 19) + (win >> (L.st & 31))];
            if (tm_trip(L, win, e, sp, [&](uint32_t bits, uint32_t) { br.pos += bits; }, TmDirectSink{coefs.data(), sp.coef_last}, store_raw))
                br.pos -= tm_overflow(L, e, sp, TmDirectSink{coefs.data(), sp.coef_last}, store_raw);
            trips++;
            if (++guard > 50000000) {
                fprintf(stderr, "slice at %u: parser did not terminate\n", units[u].off);
                return 1;
            }
        }
        tm_end(L, sp, store_raw);
        if (getenv("EFX_HARNESS_TRIPS"))
            printf("slice pic=%d code=%d type=%d trips=%ld mbs=%u bytes=%u\n", se.a, code, se.c & 15, trips - trips0, L.nmb, len);
        uint32_t nm = 0, nc = 0;
        uint32_t st = tm_finish(L, fx, tok_base, [&](uint32_t k) { return raw[k]; }, coefs.data(), reinterpret_cast<TmU4*>(recs.data()), &nm, &nc);
        if (br.pos - units[u].off * 8 > len * 8 + kTmEndBits)  // (k_parse: the slice's codes ran through the next start code)
            st |= EFX_STREAM_BAD_VLC;
        if (tm_why(L.st) == kDeadEnd) {  // (k_parse: junk between the end of the slice and the next start code)
            const uint32_t end = units[u].off + len;
            uint32_t at = br.pos >> 3;
            bool junk = false;
            if (at < end) {
                junk = (es[at] & (0xFFu >> (br.pos & 7))) != 0;
                for (at++; at < end && !junk; at++)
                    junk = es[at] != 0;
            }
            if (junk)
                st |= EFX_STREAM_SERIAL_HUNT;
        }

        if (const char* dump = getenv("EFX_HARNESS_DUMP"))
            if ((uint32_t)atol(dump) == units[u].off) {
                fprintf(stderr, "slice at %u code %d type %d r_size %u mb_limit %d: parser %u records (why %u, status %u), oracle %zu\n",
                        units[u].off, code, se.c & 15, sp.r_size, mb_limit, nm, tm_why(L.st), st, exp.size());
                for (size_t q = 0; q < exp.size(); q++)
                    fprintf(stderr, "  oracle mb %d flags %02x mv %d,%d cnt %u %u %u %u %u %u\n", exp[q].addr, exp[q].flags, exp[q].mvx,
                            exp[q].mvy, exp[q].cnt[0], exp[q].cnt[1], exp[q].cnt[2], exp[q].cnt[3], exp[q].cnt[4], exp[q].cnt[5]);
                for (uint32_t k = 0; k <= L.nmb && k < 40; k++)
                    fprintf(stderr, "  raw %u: %08x %08x %08x %08x\n", k, raw[k].x, raw[k].y, raw[k].z, raw[k].w);
            }
        if (nm != exp.size()) {
            fprintf(stderr, "slice at %u (picture %d code %d): %u macroblock records, oracle %zu (status %u, why %u)\n", units[u].off,
                    se.a, code, nm, exp.size(), st, tm_why(L.st));
            return 1;
        }
        uint32_t want_coefs = 0;
        for (const ExpMb& m : exp) {
            const MbRec& r = recs[m.addr];
            const bool skipped = m.flags & 2;
            const uint32_t want_flags = skipped ? 2u : ((m.flags & ~2u) | fx.rec_flags);
            bool same = r.epoch == epoch && r.flags == want_flags && r.mvx == (skipped ? 0 : m.mvx) && r.mvy == (skipped ? 0 : m.mvy);
            for (int k = 0; k < 6; k++)
                same = same && r.cnt[k] == m.cnt[k];
            size_t bad_k = 0;
            for (size_t k = 0; same && k < m.entries.size(); k++) {
                const uint32_t w = coefs[r.coef_base + k];
                // an intra block's first entry is the DC value pass 2 wrote; everything else is a raw stream word
                bool is_dc = false;
                if (m.flags & 1) {
                    size_t at = 0;
                    for (int b = 0; b < 6; b++) {
                        if (m.cnt[b] && at == k)
                            is_dc = true;
                        at += m.cnt[b];
                    }
                }
                const uint32_t got = is_dc ? w : (((uint32_t)tm_level(w) << 6) | (w & 63));
                same = got == m.entries[k];
                bad_k = k;
            }
            if (!same) {
                fprintf(stderr, "slice at %u (picture %d code %d) macroblock %d differs: flags %02x/%02x mv %d,%d/%d,%d cnt", units[u].off,
                        se.a, code, m.addr, r.flags, want_flags, r.mvx, r.mvy, m.mvx, m.mvy);
                for (int k = 0; k < 6; k++)
                    fprintf(stderr, " %u/%u", r.cnt[k], m.cnt[k]);
                fprintf(stderr, " (entry %zu)\n", bad_k);
                return 1;
            }
            want_coefs += (uint32_t)m.entries.size();
        }
        if (nc != want_coefs) {
            fprintf(stderr, "slice at %u: %u entries counted, oracle %u\n", units[u].off, nc, want_coefs);
            return 1;
        }
        // what the oracle says about the slice's health, against the status bits
        bool o_bad = false, o_over = false;
        for (size_t q = ev; q < e_cut; q++)
            if (g_ev[q].kind == EFXO_T_BLOCK) {
                o_bad |= g_ev[q].b == -2;
                o_over |= g_ev[q].b == -1;
            }
        if (o_over != ((st & EFX_STREAM_COEF_OVERRUN) != 0) || (o_bad && !(st & EFX_STREAM_BAD_VLC))) {
            fprintf(stderr, "slice at %u (picture %d code %d): status %u, oracle abandoned=%d bad=%d\n", units[u].off, se.a, code, st, o_over,
                    o_bad);
            return 1;
        }
        if (getenv("EFX_HARNESS_CLEAN") && st) {
            fprintf(stderr, "slice at %u (picture %d code %d): status %u on a stream declared clean\n", units[u].off, se.a, code, st);
            return 1;
        }
        slices++;
        mbs += nm;
        entries += nc;
        ev = e2;
    }
    printf("OK slices=%ld macroblocks=%ld entries=%ld trips=%ld rejected=%ld unseen=%ld phantom=%ld\n", slices, mbs, entries, trips, rejected, unseen,
           phantom + (long)(oslices.size() - os));
    return 0;
}
