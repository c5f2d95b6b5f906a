// tests/parse_harness.cpp -- TEST TOOL: runs the lane-level slice parser of k_parse
// (espflix_amd/csrc/parse_core.h, compiled here for the host) over every slice of an elementary
// stream and checks each macroblock record and coefficient entry against the parse trace of the
// test oracle (oracle/efx_oracle.c, efxo_set_trace) decoding the same stream.  CPU only: it lets the
// parser's state machine be verified without a GPU; the -m gpu tests then check the kernel built
// from the same header against the frames of the reference.
//
//   parse_harness <file> [ts]     exit 0 and "OK slices=.. macroblocks=.. entries=.." or a mismatch report
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "efx_oracle.h"
#include "parse_core.h"

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
    void advance(uint32_t n) { pos += n; }
};

struct ExpMb {
    int addr;
    uint32_t flags;
    int mvx, mvy;
    uint32_t cnt[6];
    std::vector<uint32_t> entries;
};

template <bool kAllIntra>
bool run_slice(const uint8_t* es, uint32_t off, uint32_t len, int code, const SliceParams& sp, const ParseTables& tab,
               std::vector<uint32_t>& coefs, std::vector<MbRec>& recs, uint32_t* n_coefs, uint32_t* n_mbs, uint32_t* status)
{
    HostBits br{es + off, 0};
    SliceParser<HostBits, ParseTables, kAllIntra> L;
    L.begin(br, code, off * kCoefsPerEsByte);
    long guard = 0;
    while (L.st != kLaneDone) {
        if (L.st == kLaneCoef)
            L.coef_step(br, tab, coefs.data(), sp);
        else
            L.service(br, tab, coefs.data(), recs.data(), sp);
        if (++guard > 50000000)
            return false;
    }
    (void)len;
    *n_coefs = L.n_coefs;
    *n_mbs = L.n_mbs;
    *status = L.status;
    return true;
}

}  // namespace

int main(int argc, char** argv)
{
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
    efxo_decode(in.data(), in.size(), ts ? EFXO_FMT_TS : EFXO_FMT_ES, 1, nullptr, nullptr, nullptr, 0);
    efxo_set_trace(nullptr, nullptr);

    // slice start codes in stream order (byte aligned, up to the first sequence_end_code)
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

    ParseTables* tab = new ParseTables;
    build_parse_tables(tab);
    std::vector<uint32_t> coefs(es.size() * kCoefsPerEsByte + 16);
    std::vector<MbRec> recs(kMbCount);

    size_t ev = 0;
    long slices = 0, mbs = 0, entries = 0, rejected = 0;
    const uint32_t epoch = 7;
    for (size_t u = 0; u < units.size(); u++) {
        const int code = units[u].code;
        if (code < 0x01 || code > 0xAF)
            continue;
        // the oracle's event for this slice
        while (ev < g_ev.size() && g_ev[ev].kind != EFXO_T_SLICE)
            ev++;
        if (ev >= g_ev.size()) {
            fprintf(stderr, "slice at %u: the oracle saw no more slices\n", units[u].off);
            return 1;
        }
        const Ev se = g_ev[ev++];
        if (se.b != code) {
            fprintf(stderr, "slice order differs at %u: code %02x vs oracle %02x\n", units[u].off, code, se.b);
            return 1;
        }
        const bool decoded = (se.c >> 16) & 1;
        if (se.a < 0 || !decoded || code - 2 >= kMbH) {
            rejected++;
            continue;
        }
        const uint32_t next = (u + 1 < units.size()) ? units[u + 1].off - 4 : (uint32_t)(es_len + kEsTailBytes);
        const uint32_t len = next - units[u].off;
        SliceParams sp;
        sp.coef_last = (units[u].off + len) * kCoefsPerEsByte - 1;
        sp.i_picture = (se.c & 15) == 1;
        sp.full_pel = (se.c >> 4) & 1;
        sp.r_size = (se.c >> 8) & 7;
        sp.rec_flags = se.e ? 0x80u : 0u;
        sp.epoch = epoch;
        // expected macroblocks of this slice
        std::vector<ExpMb> exp;
        std::vector<uint32_t> blk_entries;
        size_t e2 = ev;
        for (; e2 < g_ev.size() && g_ev[e2].kind != EFXO_T_SLICE; e2++) {
            const Ev& x = g_ev[e2];
            if (x.kind == EFXO_T_MB) {
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
        for (auto& r : recs)
            memset(&r, 0, sizeof r);
        uint32_t nc = 0, nm = 0, st = 0;
        const bool all_i = sp.i_picture;
        bool ok = all_i ? run_slice<true>(es.data(), units[u].off, len, code, sp, *tab, coefs, recs, &nc, &nm, &st)
                        : run_slice<false>(es.data(), units[u].off, len, code, sp, *tab, coefs, recs, &nc, &nm, &st);
        if (ok && all_i && (slices & 1))  // the generic instantiation must agree on I pictures (mixed waves use it)
            ok = run_slice<false>(es.data(), units[u].off, len, code, sp, *tab, coefs, recs, &nc, &nm, &st);
        if (!ok) {
            fprintf(stderr, "slice at %u: parser did not terminate\n", units[u].off);
            return 1;
        }
        if (nm != exp.size()) {
            fprintf(stderr, "slice at %u (picture %d code %d): %u macroblock records, oracle %zu (status %u)\n", units[u].off, se.a,
                    code, nm, exp.size(), st);
            return 1;
        }
        uint32_t want_coefs = 0;
        for (const ExpMb& m : exp) {
            const MbRec& r = recs[m.addr];
            const bool skipped = m.flags & 2;
            const uint32_t want_flags = skipped ? 2u : ((m.flags & ~2u) | sp.rec_flags);
            bool same = r.epoch == epoch && r.flags == want_flags && r.mvx == (skipped ? 0 : m.mvx) && r.mvy == (skipped ? 0 : m.mvy);
            for (int k = 0; k < 6; k++)
                same = same && r.cnt[k] == m.cnt[k];
            for (size_t k = 0; same && k < m.entries.size(); k++)
                same = coefs[r.coef_base + k] == m.entries[k];
            if (!same) {
                fprintf(stderr, "slice at %u (picture %d code %d) macroblock %d differs: flags %02x/%02x mv %d,%d/%d,%d cnt", units[u].off,
                        se.a, code, m.addr, r.flags, want_flags, r.mvx, r.mvy, m.mvx, m.mvy);
                for (int k = 0; k < 6; k++)
                    fprintf(stderr, " %u/%u", r.cnt[k], m.cnt[k]);
                fprintf(stderr, "\n");
                return 1;
            }
            want_coefs += (uint32_t)m.entries.size();
            entries += (long)m.entries.size();
        }
        if (nc != want_coefs) {
            fprintf(stderr, "slice at %u: %u coefficient entries counted, oracle %u\n", units[u].off, nc, want_coefs);
            return 1;
        }
        mbs += nm;
        slices++;
        ev = e2;
    }
    printf("OK slices=%ld macroblocks=%ld entries=%ld rejected=%ld\n", slices, mbs, entries, rejected);
    delete tab;
    return 0;
}
