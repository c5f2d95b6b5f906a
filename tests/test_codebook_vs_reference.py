"""Exhaustive check of the VLC code books in espflix_amd/csrc/mpeg1_codebook.h against the
reference's own tables, read from /root/reference/src/player.cpp at test time (build container
only): the packed binary-tree tables (player.cpp:59-116) are walked and the prefix-class DCT
tables (535-546) are re-expanded."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/player.cpp"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference sources not present")


def ours(name):
    src = open(os.path.join(ROOT, "espflix_amd", "csrc", "mpeg1_codebook.h")).read()
    body = re.search(r"static const \w+ " + name + r"\[\d+\] = \{(.*?)\};", src, re.S).group(1)
    return [tuple(int(x, 0) for x in m.split(",")) for m in re.findall(r"\{([^{}]*)\}", body)]


def ref_tree(name):
    src = open(REF).read()
    m = re.search(r"const uint32_t " + name + r"\[\d+\] = \{(.*?)\};", src, re.S)
    t = [int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]+", m.group(1))]
    out = {}

    def rec(state, code, ln):
        v = t[state]
        if (v >> 24) == 0 and state != 0:
            val = v & 0xFFFF
            out[(code, ln)] = val - 0x10000 if val >= 0x8000 else val
            return
        for bit, sh in ((0, 24), (1, 16)):
            nxt = (v >> sh) & 0xFF
            if nxt != 0xFF:
                rec(nxt, (code << 1) | bit, ln + 1)
    rec(0, 0, 0)
    return out


@pytest.mark.parametrize("ours_name,ref_name", [("kMbaCodes", "macroblock_address_increment"), ("kTypeICodes", "macroblock_type_I"),
                                                ("kTypePCodes", "macroblock_type_P"), ("kCbpCodes", "coded_block_pattern"),
                                                ("kMotionCodes", "motion_vec")])
def test_tree_books(ours_name, ref_name):
    mine = {(c, l): v for c, l, v in ours(ours_name)}
    assert mine == ref_tree(ref_name)


def test_dct_book():
    src = open(REF).read()

    def arr(name):
        body = re.search(r"const int16_t " + name + r"\[[^\]]*\] =\s*\{(.*?)\};", src, re.S).group(1)
        return [(int(a), int(b)) for a, b in re.findall(r"C\((\d+),(\d+)\)", body)] if name != "t_001" else \
            [(0, 0)] + [(int(a), int(b)) for a, b in re.findall(r"C\((\d+),(\d+)\)", body)]
    ref = {}
    ref[(0b011, 3)] = (1, 1)
    ref[(0b0100, 4)] = (0, 2)
    ref[(0b0101, 4)] = (2, 1)
    t = arr("t_001")
    for i in (1, 2, 3):
        ref[(0b00100 | i, 5)] = t[i]
    for i, rl in enumerate(arr("t_00100")):
        ref[(0b00100000 | i, 8)] = rl
    for i, rl in enumerate(arr("t_0001")):
        ref[(0b000100 | i, 6)] = rl
    for i, rl in enumerate(arr("t_00001")):
        ref[(0b0000100 | i, 7)] = rl
    for i, rl in enumerate(arr("t_0000001")):
        ref[(0b0000001000 | i, 10)] = rl
    big = arr("t_0000X")
    for z in range(5):
        for i in range(16):
            ref[((1 << 4) | i, 12 + z)] = big[z * 16 + i]
    mine = {(c, l): (r, lv) for c, l, r, lv in ours("kDctCodes")}
    assert mine == ref
