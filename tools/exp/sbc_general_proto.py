#!/usr/bin/env python3
"""CPU prototype of the frame-parallel SBC decoder's GENERAL path (k_sbc_frames -> k_sbc_plan -> k_sbc_gen in
espflix_amd/csrc/k_sbc.hip), checked against the test oracle on mutated streams.  It restates, chunk by chunk and with the
same tables (frame info, frame plan, slots, row map), what the kernels compute -- the design was settled here before a
GPU minute was spent on it.  Run from the repo root:  python tools/exp/sbc_general_proto.py [seeds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import oracle  # noqa: E402

CHUNK = 8
OFFSET8 = [[-2, 0, 0, 0, 0, 0, 0, 1], [-3, 0, 0, 0, 0, 0, 1, 2], [-4, 0, 0, 0, 0, 0, 1, 2], [-4, 0, 0, 0, 0, 0, 1, 2]]


def tables():
    import ctypes as C
    L = oracle.lib()
    syn, proto = np.zeros(128, np.int32), np.zeros(80, np.int32)
    L.efxo_sbc_tables.argtypes = [C.c_void_p, C.c_void_p]
    L.efxo_sbc_tables(syn.ctypes.data, proto.ctypes.data)
    return syn.astype(np.int64), proto.astype(np.int64)


SYN, PROTO = tables()


def wrap32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def bit_allocation(freq, alloc, bitpool, scale):
    bitneed = []
    for sb in range(8):
        s = scale[sb]
        if alloc:
            need = s
        elif s == 0:
            need = -5
        else:
            l = s - OFFSET8[freq][sb]
            need = l // 2 if l > 0 else l
        bitneed.append(need)
    max_bitneed = max(0, max(bitneed))
    bitcount, slicecount, bitslice = 0, 0, max_bitneed + 1
    while True:
        bitslice -= 1
        bitcount += slicecount
        slicecount = 0
        for sb in range(8):
            if bitslice + 1 < bitneed[sb] < bitslice + 16:
                slicecount += 1
            elif bitneed[sb] == bitslice + 1:
                slicecount += 2
        if not bitcount + slicecount < bitpool:
            break
    if bitcount + slicecount == bitpool:
        bitcount += slicecount
        bitslice -= 1
    bits = [0] * 8
    for sb in range(8):
        if bitneed[sb] >= bitslice + 2:
            bits[sb] = min(bitneed[sb] - bitslice, 16)
    for sb in range(8):
        if bitcount >= bitpool:
            break
        if 2 <= bits[sb] < 16:
            bits[sb] += 1
            bitcount += 1
        elif bitneed[sb] == bitslice + 1 and bitpool > bitcount + 1:
            bits[sb] = 2
            bitcount += 2
    for sb in range(8):
        if bitcount >= bitpool:
            break
        if bits[sb] < 16:
            bits[sb] += 1
            bitcount += 1
    return bits


def iq_magic(bits):
    """Granlund-Montgomery round-up constants of the divisor 2^bits - 1 (what build_sbc_tables() stores)."""
    d = (1 << bits) - 1
    if d == 1:
        return 1, 0, 0
    l = bits
    m = ((1 << 32) * ((1 << l) - d)) // d + 1
    return m & 0xFFFFFFFF, 1, l - 1


def iquant(v, bits, scale):
    n = (((v << 1) | 1) << scale) & 0xFFFFFFFF
    neg = n >> 31
    a = (-n) & 0xFFFFFFFF if neg else n
    m, sh1, sh2 = iq_magic(bits)
    t1 = (m * a) >> 32
    q = (t1 + (((a - t1) & 0xFFFFFFFF) >> sh1)) >> sh2
    assert q == a // ((1 << bits) - 1)
    return (-q if neg else q) - (1 << scale)


class State:
    def __init__(self):
        self.hist = np.zeros((2, 9, 16), np.int64)
        self.sb = np.zeros((16, 2, 8), np.int64)
        self.freq = self.blocks = self.channels = self.mode = self.alloc = self.subbands = self.bitpool = 0


def peek(data, limit, bitpos, n):
    b = bitpos >> 3
    w = 0
    for k in range(3):
        w = (w << 8) | (int(data[b + k]) if b + k < limit else 0)
    return (w >> (24 - (bitpos & 7) - n)) & ((1 << n) - 1)


def k_frames(data, fb, n_frames, limit):
    """per real frame: sync, ok, h1, bitpool, bits[2][8], prefix[2][8], per_block, framelen"""
    info = []
    for f in range(n_frames):
        d = data[f * fb:]
        avail = limit - f * fb
        e = dict(sync=False, ok=False, h1=0, bitpool=0, bits=[[0] * 8, [0] * 8], prefix=[[0] * 8, [0] * 8], per_block=0, framelen=-1,
                 scale=[[0] * 8, [0] * 8])
        if fb >= 4 and d[0] == 0x9C:
            h1, bp = int(d[1]), int(d[2])
            mode = (h1 >> 2) & 3
            e.update(sync=True, h1=h1, bitpool=bp)
            e["ok"] = mode != 3 and (h1 & 1) == 1 and bp <= 128
            if e["ok"]:
                ch = 2 if mode else 1
                acc = 0
                for c in range(ch):
                    sc = []
                    for j in range(8):
                        i = 4 + ((c * 8 + j) >> 1)
                        a = int(d[i]) if i < avail else 0
                        sc.append((a & 0xF) if (c * 8 + j) & 1 else a >> 4)
                    e["scale"][c] = sc
                    b = bit_allocation((h1 >> 6) & 3, (h1 >> 1) & 1, bp, sc)
                    for j in range(8):
                        e["bits"][c][j] = b[j]
                        e["prefix"][c][j] = acc
                        acc += b[j]
                e["per_block"] = acc
                blocks = 4 * (((h1 >> 4) & 3) + 1)
                e["framelen"] = 4 + ch * 4 + (blocks * acc + 7) // 8
        info.append(e)
    return info


def geom_of_h1(h1):
    mode = (h1 >> 2) & 3
    return 4 * (((h1 >> 4) & 3) + 1), 2 if mode else 1, (h1 & 1) == 0  # blocks, channels, four subbands


def k_plan(info, n_frames, probe, st):
    """per virtual frame v (0 .. F-1; real frame max(v - probe, 0)): the scans"""
    F = n_frames + probe
    g_state = (min(st.blocks, 16), min(st.channels, 2), st.subbands == 4)
    plan = []
    gsrc, src, back = -1, [-1] * 8, [-1, -1]
    vb, pcm_off = [0, 0], 0
    rets = []
    for v in range(F):
        e = info[max(v - probe, 0)]
        if e["sync"]:
            gsrc = v
        blocks, channels, four = geom_of_h1(info[max(gsrc - probe, 0)]["h1"]) if gsrc >= 0 else g_state
        if e["ok"]:
            hb, hc, _ = geom_of_h1(e["h1"])
            for q in range(4):
                for c in range(2):
                    if hb > 4 * q and hc > c:
                        src[q * 2 + c] = v
        synth = not four
        nb = [blocks if synth and channels > c else 0 for c in range(2)]
        pcm = blocks * 8 * channels if synth else 0
        plan.append(dict(src=list(src), pcm_off=pcm_off, vb=list(vb), back=list(back), gsrc=gsrc, blocks=blocks, channels=channels, synth=synth))
        for c in range(2):
            if nb[c]:
                back[c] = v
            vb[c] += nb[c]
        if not (probe and v == 0):
            pcm_off += pcm
            rets.append((e["framelen"] if e["ok"] else -1, pcm * 2))
    return plan, pcm_off, rets, list(vb), list(back)


def k_gen(data, fb, n_frames, probe, st, info, plan, limit, v0, out):
    """one workgroup: virtual frames v0 .. v1-1.  Returns the new state when it holds the last frame."""
    F = n_frames + probe
    v1 = min(F, v0 + CHUNK)
    # slots: per channel up to three earlier frames that hold the nine rows before the chunk, then the chunk's frames
    slots = []
    base = [0, 0]
    for c in range(2):
        need, g = 9, plan[v0]["back"][c]
        hops = 0
        while need > 0 and g >= 0:
            assert hops < 3
            slots.append(g)
            need -= plan[g]["blocks"]  # (a frame on channel c's chain has rows on it)
            g = plan[g]["back"][c]
            hops += 1
        while hops < 3:
            slots.append(-1)
            hops += 1
        base[c] = plan[v0]["vb"][c] - 9
    slots += list(range(v0, v1))
    # samples of every slot: the sb_sample array as it stood after that frame's get_samples()
    sb = np.zeros((len(slots), 16, 2, 8), np.int64)
    for j, u in enumerate(slots):
        if u < 0:
            continue
        for blk in range(16):
            for c in range(2):
                h = plan[u]["src"][(blk >> 2) * 2 + c]
                if h < 0:
                    sb[j, blk, c] = st.sb[blk, c]
                    continue
                fh = max(h - probe, 0)
                e = info[fh]
                hb, hc, _ = geom_of_h1(e["h1"])
                assert e["ok"] and blk < hb and c < hc
                doff = 4 + hc * 4
                for s in range(8):
                    b = e["bits"][c][s]
                    if b:
                        bitpos = doff * 8 + blk * e["per_block"] + e["prefix"][c][s]
                        sb[j, blk, c, s] = iquant(peek(data[fh * fb:], limit - fh * fb, bitpos, b), b, e["scale"][c][s])
    # row maps
    n_t = [0, 0]
    rows = np.zeros((2, 9 + CHUNK * 16, 16), np.int64)
    rowmap = [[None] * (9 + CHUNK * 16) for _ in range(2)]
    end_vb = [0, 0]
    for c in range(2):
        last = plan[v1 - 1]
        end_vb[c] = last["vb"][c] + (last["blocks"] if last["synth"] and last["channels"] > c else 0)
        n_t[c] = 9 + end_vb[c] - plan[v0]["vb"][c]
        for j, u in enumerate(slots):
            if u < 0:
                continue
            if j < 6 and j // 3 != c:
                continue  # (the other channel's earlier frames)
            p = plan[u]
            if not (p["synth"] and p["channels"] > c):
                continue
            for blk in range(p["blocks"]):
                t = p["vb"][c] + blk - base[c]
                if 0 <= t < n_t[c]:
                    rowmap[c][t] = (j, blk)
        for t in range(9):
            T = base[c] + t
            if T < 0:
                assert rowmap[c][t] is None
                rows[c, t] = st.hist[c, 9 + T]
            else:
                assert rowmap[c][t] is not None, (c, t, T, slots, v0)
        for t in range(n_t[c]):
            if rowmap[c][t] is None:
                continue
            j, blk = rowmap[c][t]
            for o in range(16):
                acc = 0
                for k in range(8):
                    acc += int(SYN[o * 8 + k]) * int(sb[j, blk, c, k])
                rows[c, t, o] = wrap32(acc) >> 15
    # windowing
    for v in range(v0, v1):
        p = plan[v]
        if not p["synth"] or (probe and v == 0):
            continue
        for c in range(p["channels"]):
            for blk in range(p["blocks"]):
                t = p["vb"][c] + blk - base[c]
                for o in range(8):
                    acc = 0
                    for j in range(0, 10, 2):
                        acc += int(rows[c, t - j, o]) * int(PROTO[o * 10 + j])
                        acc += int(rows[c, t - j - 1, o + 8]) * int(PROTO[o * 10 + j + 1])
                    val = wrap32(acc) >> 15
                    val = max(-0x7FFF, min(0x7FFF, val))
                    out[p["pcm_off"] + c * p["blocks"] * 8 + blk * 8 + o] = val
    if v1 != F:
        return None
    ns = State()
    ns.sb = sb[len(slots) - 1].copy()
    for c in range(2):
        ns.hist[c] = rows[c, n_t[c] - 9:n_t[c]]
    g = plan[F - 1]["gsrc"]
    if g >= 0:
        h1 = info[max(g - probe, 0)]["h1"]
        ns.freq, ns.mode, ns.alloc, ns.bitpool = (h1 >> 6) & 3, (h1 >> 2) & 3, (h1 >> 1) & 1, info[max(g - probe, 0)]["bitpool"]
        ns.blocks, ns.channels, four = geom_of_h1(h1)
        ns.subbands = 4 if four else 8
    else:
        ns.freq, ns.mode, ns.alloc, ns.bitpool = st.freq, st.mode, st.alloc, st.bitpool
        ns.blocks, ns.channels, ns.subbands = min(st.blocks, 16), min(st.channels, 2), 4 if st.subbands == 4 else 8
    return ns


def decode_call(data, fb, n_frames, probe, st, limit):
    """(limit: the kernels use n_frames * fb -- bytes beyond a call's frames read as zero; the prototype's three-call runs pass
    the rest of the stream instead, so that frames that run past the stated frame size compare with the one-call oracle)"""
    data = np.concatenate([data[:limit], np.zeros(2048, np.uint8)])
    info = k_frames(data, fb, n_frames, limit)
    plan, total, rets, total_vb, _ = k_plan(info, n_frames, probe, st)
    out = np.zeros(total, np.int64)
    ns = st
    F = n_frames + probe
    order = list(range(0, F, CHUNK))
    np.random.default_rng(n_frames).shuffle(order)  # workgroups run in any order
    for v0 in order:
        r = k_gen(data, fb, n_frames, probe, st, info, plan, limit, v0, out)
        if r is not None:
            ns = r
    return out.astype(np.int16), rets, ns


def mutate(rng, fr, fb, n):
    fr = fr.reshape(n, fb).copy()
    for _ in range(rng.integers(1, 8)):
        f = int(rng.integers(0, n))
        kind = int(rng.integers(0, 8))
        if kind == 0:
            fr[f, 0] = 0x9D
        elif kind == 1:
            fr[f, 1] |= 0x0C
        elif kind == 2:
            fr[f, 1] &= 0xFE
        elif kind == 3:
            fr[f, 2] = 200
        elif kind == 4:  # another block count
            fr[f, 1] = (fr[f, 1] & 0xCF) | (int(rng.integers(0, 4)) << 4)
        elif kind == 5:  # mono <-> dual / stereo
            fr[f, 1] = (fr[f, 1] & 0xF3) | (int(rng.integers(0, 3)) << 2)
        elif kind == 6:  # a run of bad sync bytes
            fr[f:f + int(rng.integers(2, 12)), 0] = 0
        else:
            fr[f, 2] = int(rng.integers(2, 129))
    return fr.reshape(-1)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = 0
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        name, kw, _, _ = common.SBC_CASES[seed % len(common.SBC_CASES)]
        n = int(rng.integers(3, 40))
        ch = 1 if kw["mode"] == 0 else 2
        fb = common.sbc_frame_bytes(kw["blocks"], ch, kw["bitpool"])
        fr = common.sbc_frames(seed, n, **kw)
        if seed % 5:
            fr = mutate(rng, fr, fb, n)
        probe = seed % 3 == 0
        want, wret = oracle.sbc_decode(fr, fb, probe)
        if probe:
            want, wret = want[wret[0][1] // 2:], wret[1:]
        # one call, then the same frames in three calls with the state carried over
        for calls in (1, 3):
            st = State()
            got, rets = [], []
            per = (n + calls - 1) // calls
            for c in range(calls):
                f0, f1 = c * per, min(n, (c + 1) * per)
                if f1 <= f0:
                    continue
                pcm, r, st = decode_call(fr[f0 * fb:], fb, f1 - f0, probe and c == 0, st, (n - f0) * fb)
                got.append(pcm)
                rets += r
            got = np.concatenate(got)
            ok = np.array_equal(got, want) and rets == wret
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, name, "frames", n, "probe", probe, "calls", calls, len(got), len(want),
                      [i for i, (a, b) in enumerate(zip(rets, wret)) if a != b][:5])
    print("seeds", seeds, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
