"""Shared deterministic inputs for the tests and for tests/golden/make_golden.py."""
import numpy as np

FRAME_BYTES = 101376
SYN_FLAGS = [0, 1, 2, 4, 8, 16, 64, 128, 256, 260]   # generator flavours (espflix_amd.gen.FLAG_*); 64: every escape form of
                                           # player.cpp:1092-1099, 128: ignored picture types, user data / extension units,
                                           # 256 / 260: extra_information_slice (player.cpp:1261-1262) in 12- / 5-slice pictures
SYN_IDS = [0, 1, 7]               # 7: full_pel_forward = 1, odd ids: forward_f_code = 2


def lcg_frames(seed: int = 12345) -> np.ndarray:
    """Two frames filled by the LCG of SURVEY.md section 8c: s = s*1664525 + 1013904223,
    byte = min(s >> 24, 248), strip by strip (rows of 528 bytes incl. chroma)."""
    out = np.empty(2 * FRAME_BYTES, dtype=np.uint8)
    s = np.uint64(seed)
    vals = np.empty(out.size, dtype=np.uint64)
    x = int(seed)
    for i in range(out.size):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = x >> 24
    return np.minimum(vals, 248).astype(np.uint8)


def random_frames(seed: int) -> np.ndarray:
    """Two frames of full-range random bytes (exercises the blitter's carries)."""
    return np.random.default_rng(seed).integers(0, 256, size=2 * FRAME_BYTES, dtype=np.uint8)


def pdm_pcm(k: int = 0, calls: int = 40) -> np.ndarray:
    """x[n] = 8000 sin(2 pi (220+k) n / 48000), plus two calls of
    full-scale noise."""
    n = np.arange(128 * calls)
    x = np.round(8000 * np.sin(2 * np.pi * (220 + k) * n / 48000)).astype(np.int16)
    rng = np.random.default_rng(100 + k)
    x[128 * 10:128 * 12] = rng.integers(-32768, 32767, 256)
    return x


def fnv_bytes(a: np.ndarray) -> int:
    import oracle
    return oracle.fnv1a64(np.ascontiguousarray(a).view(np.uint8).reshape(-1))


# ---- transport-stream packetiser (test input only) -------------------------------------------
def _pts_field(prefix: int, pts: int) -> bytes:
    a = (prefix << 4) | (((pts >> 30) & 7) << 1) | 1
    b = (((pts >> 15) & 0x7FFF) << 1) | 1
    c = ((pts & 0x7FFF) << 1) | 1
    return bytes([a, b >> 8, b & 0xFF, c >> 8, c & 0xFF])


def ts_packet(pid: int, payload: bytes, pusi: bool = False, cc: int = 0, afc_only: bool = False) -> bytes:
    """One 188-byte packet; short payloads are padded with an adaptation field of stuffing."""
    assert len(payload) <= 184
    hdr = bytes([0x47, (0x40 if pusi else 0) | (pid >> 8), pid & 0xFF])
    if afc_only:  # adaptation field only, no payload flag
        return hdr + bytes([0x20 | (cc & 15), 183, 0x00]) + b"\xFF" * 182
    if len(payload) == 184:
        return hdr + bytes([0x10 | (cc & 15)]) + payload
    L = 183 - len(payload)
    af = bytes([L]) + (bytes([0x00]) + b"\xFF" * (L - 1) if L else b"")
    return hdr + bytes([0x30 | (cc & 15)]) + af + payload


def pes_header(pts: int | None, dts: bool = False, stuffing: int = 0) -> bytes:
    """Video PES header as MpegDecoder::demux reads it (player.cpp:381-405): 00 00 01 E0, length
    0, flags, header_data_length, [PTS], [DTS], stuffing."""
    opt = b""
    flags2 = 0
    if pts is not None:
        flags2 = 0xC0 if dts else 0x80
        opt += _pts_field(3 if dts else 2, pts)
        if dts:
            opt += _pts_field(1, max(pts - 3003, 0))
    opt += b"\xFF" * stuffing
    return bytes([0, 0, 1, 0xE0, 0, 0, 0x80, flags2, len(opt)]) + opt


def packetize(es: bytes, pes_starts, rng, noise: bool = True, min_payload: int = 1) -> bytes:
    """Wrap an elementary stream into a transport stream the way a hostile-but-legal muxer might:
    pes_starts = [(es_offset, pts or None, dts, stuffing)], sorted; payload sizes are random;
    null / audio-PID / adaptation-only packets are interleaved when `noise`."""
    out = bytearray()
    cc = 0
    bounds = [p[0] for p in pes_starts] + [len(es)]
    assert bounds[0] == 0
    for k, (off, pts, dts, stuffing) in enumerate(pes_starts):
        chunk = es[off:bounds[k + 1]]
        first = True
        pos = 0
        while first or pos < len(chunk):
            head = pes_header(pts, dts, stuffing) if first else b""
            room = 184 - len(head)
            n = min(len(chunk) - pos, int(rng.integers(min_payload, room + 1)) if rng.integers(0, 3) else room)
            out += ts_packet(0x100, head + chunk[pos:pos + n], pusi=first, cc=cc)
            cc += 1
            pos += n
            first = False
            if noise:
                r = int(rng.integers(0, 12))
                if r == 0:
                    out += ts_packet(0x1FFF, bytes(184))
                elif r == 1:  # an audio PES the video path must ignore
                    out += ts_packet(0x102, pes_header(int(rng.integers(0, 1 << 33))) + bytes(rng.integers(0, 256, 40, dtype=np.uint8)), pusi=True)
                elif r == 2:
                    out += ts_packet(0x100, b"", afc_only=True, cc=cc)
                elif r == 3:
                    out += ts_packet(0x101, bytes(rng.integers(0, 256, 184, dtype=np.uint8)))
    return bytes(out)


def picture_offsets(es: bytes):
    """Offsets of every picture_start_code (00 00 01 00) in an elementary stream."""
    a = np.frombuffer(es, dtype=np.uint8)
    m = (a[:-3] == 0) & (a[1:-2] == 0) & (a[2:-1] == 1) & (a[3:] == 0)
    return [int(i) for i in np.nonzero(m)[0]]


def hostile_ts(es: bytes, seed: int, noise: bool = True) -> bytes:
    """PES boundaries placed a few bytes either side of the picture start codes (the PTS latch
    depends on the bit reader's look-ahead), PTS present / absent / with DTS, header stuffing."""
    rng = np.random.default_rng(seed)
    starts = [(0, 129003, False, 0)]
    for i, p in enumerate(picture_offsets(es)):
        off = p + int(rng.integers(-3, 9))
        if off <= starts[-1][0] or off >= len(es):
            continue
        kind = int(rng.integers(0, 6))
        pts = None if kind == 0 else 200000 + 3003 * i + int(rng.integers(0, 100))
        starts.append((off, pts, kind == 1, int(rng.integers(0, 4)) if kind == 2 else 0))
    return packetize(es, starts, rng, noise=noise)


def late_pts_ts(es: bytes, first_with_pts: int) -> bytes:
    """One PES per picture, the first `first_with_pts` of them WITHOUT a PTS: the reference neither pushes nor
    swaps its buffers until a picture header has latched one (flush_picture, player.cpp:692-702)."""
    offs = picture_offsets(es)
    # a PES starts at the first header of each picture's group (sequence / GOP headers travel with their picture)
    starts = [0] + offs[1:]
    pes = [(o, None if i < first_with_pts else 129003 + 3003 * i, False, 0) for i, o in enumerate(starts)]
    return packetize(es, pes, np.random.default_rng(7), noise=False, min_payload=100)


# ---- display-state cases shared by the oracle-vs-reference, golden and GPU tests ---------------
EASE = [0, 8, 16, 24, 48, 72, 104, 136, 176, 216, 248, 280, 304, 328, 336, 344]   # _easd, video.cpp:1076


def overlay_bytes(seed: int) -> np.ndarray:
    """An 80 x 16 overlay (_video_composite): random glyph-like bytes incl. 0 and 255."""
    rng = np.random.default_rng(seed)
    o = rng.integers(0, 256, 1280, dtype=np.uint8)
    o[rng.integers(0, 1280, 200)] = 0
    o[rng.integers(0, 1280, 50)] = 255
    return o


# (name, front, hscroll per field or None, overlay seed or None, blend, progress)
DISPLAY_CASES = [
    ("slide_in", 0, [0] + [-e for e in EASE[::-1]], None, 0, 0),        # animate() with _animate_index = -16
    ("slide_out", 1, [0] + EASE[::-1], None, 0, 0),                       # _animate_index = 16
    ("overlay_full", 0, None, 5, -1, 100),
    ("overlay_fade", 1, None, 6, 34, 239),                                # 34, 33, 32 full, then 31.. fading
    ("overlay_bar_only", 0, None, None, 3, 1),
    ("both", 0, [8, -8, 344, -344, 176, -176], 7, 40, 240),
]


# ---- SBC frames (test input only) ------------------------------------------------------------
def sbc_frame_bytes(blocks: int, channels: int, bitpool: int) -> int:
    return 4 + channels * 4 + (blocks * channels * bitpool + 7) // 8


def sbc_frames(seed: int, n: int, freq: int = 3, blocks: int = 16, mode: int = 0, alloc: int = 0, bitpool: int = 28,
               max_scale: int = 12) -> np.ndarray:
    """n syntactically valid 8-subband SBC frames (A2DP frame layout as sbc_decoder.cpp:276-344
    reads it): header 9C | freq blocks mode alloc 1 | bitpool | crc (ignored by the reference),
    4-bit scale factors, then random sample bits.  Any bit pattern is a valid frame."""
    rng = np.random.default_rng(seed)
    channels = 1 if mode == 0 else 2
    fb = sbc_frame_bytes(blocks, channels, bitpool)
    out = np.zeros((n, fb), dtype=np.uint8)
    out[:, 0] = 0x9C
    out[:, 1] = (freq << 6) | ({4: 0, 8: 1, 12: 2, 16: 3}[blocks] << 4) | (mode << 2) | (alloc << 1) | 1
    out[:, 2] = bitpool
    out[:, 3] = rng.integers(0, 256, n)
    sf = rng.integers(0, max_scale + 1, (n, channels * 8))
    out[:, 4:4 + channels * 4] = (sf[:, 0::2] << 4) | sf[:, 1::2]
    out[:, 4 + channels * 4:] = rng.integers(0, 256, (n, fb - 4 - channels * 4))
    return out.reshape(-1)


def sbc_mutate(rng, frames: np.ndarray, fb: int, n: int, hits: int | None = None) -> np.ndarray:
    """Frames the reference rejects or decodes under another geometry, sprinkled over a valid stream: a bad sync byte
    (the previous samples are synthesised again), joint stereo (the geometry moves, stale samples), a 4-subband header
    (nothing is synthesised until a good frame), an impossible bitpool, another block count or channel count (the PCM
    size of a frame changes mid-stream), runs of bad sync bytes, another bitpool (the frame runs past -- or stops short
    of -- the frame size)."""
    fr = frames.reshape(n, fb).copy()
    for _ in range(int(rng.integers(1, 8)) if hits is None else hits):
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
        elif kind == 4:
            fr[f, 1] = (fr[f, 1] & 0xCF) | (int(rng.integers(0, 4)) << 4)
        elif kind == 5:
            fr[f, 1] = (fr[f, 1] & 0xF3) | (int(rng.integers(0, 3)) << 2)
        elif kind == 6:
            fr[f:f + int(rng.integers(2, 12)), 0] = 0
        else:
            fr[f, 2] = int(rng.integers(2, 129))
    return fr.reshape(-1)


# (name, kwargs, frames, probe)
SBC_CASES = [
    ("espflix_mono_48k_bp28", dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28), 40, True),
    ("mono_snr_32k_bp19_b12", dict(freq=1, blocks=12, mode=0, alloc=1, bitpool=19), 30, False),
    ("dual_loud_44k_bp35_b8", dict(freq=2, blocks=8, mode=1, alloc=0, bitpool=35), 30, False),
    ("stereo_snr_16k_bp53_b4", dict(freq=0, blocks=4, mode=2, alloc=1, bitpool=53), 50, False),
    ("mono_bp2", dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=2), 12, False),
    ("mono_bp128_loud", dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=128, max_scale=15), 12, False),
    ("dual_bp128_snr", dict(freq=3, blocks=16, mode=1, alloc=1, bitpool=128, max_scale=15), 8, False),
]


def seed_of(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode()) & 0xFFFF


CLIP_SBC_FRAME_BYTES = {"splash": 64, "vmedia": 48}


def index_titles():
    """(name, [main, fwd, rwd] transport streams) for the trick-play index tests: synthetic titles
    with a sequence header per GOP (GOP 12 main, GOP 3 trick streams as the indexer's ffmpeg
    recipe makes them, indexer.cpp:292-295)."""
    from espflix_amd import gen
    out = []
    for t in range(2):
        main = gen.Batch(300 + t, 1, 96 + 24 * t, 12, 0).ts(0)
        fwd = gen.Batch(310 + t, 1, 24, 3, 0).ts(0)
        rwd = gen.Batch(320 + t, 1, 21, 3, 0).ts(0)
        out.append((f"title{t}", [main, fwd, rwd]))
    return out


def index_queries(hdr_first: int, hdr_last: int):
    """(pts, speed) probes around and beyond the indexed range."""
    q = []
    for speed in (0, 1, -1):
        for pts in [hdr_first - 5000, hdr_first, hdr_first + 1, hdr_first + 7499, hdr_first + 7500, (hdr_first + hdr_last) // 2,
                    hdr_last - 1, hdr_last, hdr_last + 90000, 0, 1 << 33]:
            q.append((int(pts), speed))
    return q


def hostile_audio_packets(seed: int):
    """Audio PES packets (PID 0x101 / 0x102) with and without PTS -- the gate of player.cpp:421-433
    opens and closes -- split over transport packets of random payload size."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(60):
        pid = 0x102 if k % 3 else 0x101
        kind = int(rng.integers(0, 4))   # 0: PES without PTS (closes the gate)
        body = bytes(rng.integers(0, 256, int(rng.integers(20, 500)), dtype=np.uint8))
        head = pes_header(None if kind == 0 else int(rng.integers(0, 1 << 33)), dts=kind == 2, stuffing=3 if kind == 3 else 0)
        pos, first = 0, True
        while first or pos < len(body):
            h = head if first else b""
            n = min(len(body) - pos, int(rng.integers(1, 184 - len(h) + 1)))
            out.append(ts_packet(pid, h + body[pos:pos + n], pusi=first))
            pos += n
            first = False
    return out


def interleave_audio(video_ts: bytes, seed: int) -> bytes:
    """A valid video transport stream with hostile audio packets (and a few null packets) mixed in."""
    rng = np.random.default_rng(seed + 1000)
    v = [video_ts[i:i + 188] for i in range(0, len(video_ts) - 187, 188)]
    a = hostile_audio_packets(seed)
    out = bytearray()
    ia = 0
    for pkt in v:
        out += pkt
        while ia < len(a) and rng.integers(0, 3) == 0:
            out += a[ia]
            ia += 1
        if rng.integers(0, 9) == 0:
            out += ts_packet(0x1FFF, bytes(184))
    for pkt in a[ia:]:
        out += pkt
    return bytes(out)


# ---- hand-built P pictures: macroblock_stuffing and macroblock_escape in every number (test input only) ----------------
class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, bits: int):
        for i in range(bits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                self.acc, self.n = 0, 0

    def align(self):
        while self.n:
            self.put(0, 1)

    def start_code(self, code: int):
        self.align()
        self.out += bytes([0, 0, 1, code])


STUFFING_COUNTS = [0, 1, 7, 8, 15, 16, 17, 40, 63, 64, 65, 127, 128, 129, 255, 256, 300]


def stuffing_es() -> bytes:
    """One generator I picture, then three P pictures written bit by bit (ISO 11172-2 2.4.2.7, 2.4.3.6 as
    MpegDecoder::slice reads them, player.cpp:1264-1316):
      picture 1  twelve slices of 22 "motion only" macroblocks with small vectors in the interior; in front of chosen
                 macroblocks -- the first of a slice included -- 1 ... 300 macroblock_stuffing codes (the reference loops
                 over them, player.cpp:1268-1270; a decoder that counts them must not let the count carry into a live field);
      picture 2  ONE slice for the whole picture: address increments of 68 (escape, stuffing, 1: after an escape the
                 reference's second loop counts a stuffing code as 34, player.cpp:1271-1274), 134 (four escapes + 2), then ones;
      picture 3  twelve slices again, every macroblock skipped except the first and the last of a slice (increment 21).
    All vectors keep the 17 x 17 fetch inside the picture (SURVEY 8a stream constraints)."""
    from espflix_amd import gen
    head = gen.Batch(0, 1, 1, 12, 0).es(0).tobytes()
    b = _Bits()
    rng = np.random.default_rng(11)
    MC = (0b001, 3)      # macroblock_type P: motion forward, no pattern
    MV = {0: (0b1, 1), 1: (0b010, 3), -1: (0b011, 3), 2: (0b0010, 4), -2: (0b0011, 4)}
    INC = {1: (0b1, 1), 2: (0b011, 3), 3: (0b010, 3), 21: (0b0000010010, 10)}  # table B-1 (21: 0000 0100 10)
    ESC, STUFF = (0b00000001000, 11), (0b00000001111, 11)

    def picture_header(tref):
        b.start_code(0x00)
        b.put(tref, 10)
        b.put(2, 3)        # P
        b.put(0xFFFF, 16)  # vbv_delay
        b.put(0, 1)        # full_pel_forward_vector
        b.put(1, 3)        # forward_f_code 1
        b.put(0, 1)        # extra_bit_picture

    def mb(inc_codes, dh, dv):
        for c in inc_codes:
            b.put(*c)
        b.put(*MC)
        b.put(*MV[dh])
        b.put(*MV[dv])

    # picture 1
    picture_header(1)
    counts = list(STUFFING_COUNTS)
    for row in range(12):
        b.start_code(row + 1)
        b.put(8, 5)
        b.put(0, 1)
        # horizontal deltas: columns 1..10 random, 11..20 undo them (the vector is 0 again at the right edge)
        dh = [0] * 22
        dv = [0] * 22
        for c in range(1, 11):
            dh[c] = int(rng.integers(-2, 3))
            dh[21 - c] = -dh[c]
            if 1 <= row <= 10:
                dv[c] = int(rng.integers(-2, 3))
                dv[21 - c] = -dv[c]
        for col in range(22):
            pre = []
            if (col * 5 + row) % 7 == 0 or (row == 3 and col == 0):
                pre = [STUFF] * counts[(row * 22 + col) % len(counts)]
            mb(pre + [INC[1]], dh[col], dv[col])
    # picture 2: one slice for the whole picture
    picture_header(2)
    b.start_code(1)
    b.put(6, 5)
    b.put(0, 1)
    mb([INC[1]], 0, 0)                 # address 0
    mb([ESC, STUFF], 0, 0)             # + 33 + 34 = 67: after an escape a stuffing code IS the increment 34
    mb([ESC] * 4 + [INC[2]], 0, 0)     # + 134 = 201
    for _ in range(62):
        mb([STUFF] * 2 + [INC[1]], 0, 0)   # ... 263
    # picture 3: first and last macroblock of every row coded, the twenty between them skipped
    picture_header(3)
    for row in range(12):
        b.start_code(row + 1)
        b.put(10, 5)
        b.put(0, 1)
        mb([INC[1]], 0, 0)
        mb([STUFF] * (row * 13) + [INC[21]], 0, 0)
    b.align()
    return head + bytes(b.out)


def one_pes_per_picture(es: bytes) -> bytes:
    """Transport stream with one PES (and PTS) per picture, the first PES also carrying the headers in front of it."""
    return late_pts_ts(es, 0)
