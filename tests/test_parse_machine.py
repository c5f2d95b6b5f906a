"""k_parse's slice parser on the host (no GPU): tests/_build/parse_harness compiles espflix_amd/csrc/parse_tm.h -- the token
machine's two passes -- and the tables of espflix_amd/csrc/efx_tables.cpp for the CPU, parses every slice of a stream the
way a lane of the kernel does, and compares each macroblock record (address, type flags, motion vector, coefficient
counts) and each coefficient entry with the parse trace of the test oracle decoding the same stream (the oracle itself is
pinned to the reference by tests/test_oracle_golden.py / test_oracle_vs_ref.py).  The -m gpu tests then check the kernel
built from the same header against whole frames."""
import os
import re
import subprocess

import numpy as np
import pytest

import common
from espflix_amd import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "_build", "parse_harness")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(HARNESS):
        subprocess.run(["make", "-C", ROOT, "harness"], check=True, capture_output=True)
    return HARNESS


def run(harness, path, *args):
    p = subprocess.run([harness, path, *args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    m = re.match(r"OK slices=(\d+) macroblocks=(\d+) entries=(\d+) trips=(\d+) rejected=(\d+) unseen=(\d+) phantom=(\d+)", p.stdout)
    assert m, p.stdout
    return dict(zip(("slices", "macroblocks", "entries", "trips", "rejected", "unseen", "phantom"), map(int, m.groups())))


@pytest.mark.parametrize("flags", common.SYN_FLAGS + [36, 68, 132])
def test_synthetic_streams(flags, harness, tmp_path):
    """Every generator flavour (escape levels, ignored picture types, extra_information_slice, 5-slice pictures, ...)."""
    b = gen.Batch(0, 8, 12, 12, flags)
    for k in common.SYN_IDS:
        f = tmp_path / f"s{flags}_{k}.es"
        b.es(k).tofile(f)
        r = run(harness, str(f))
        per_picture = 5 if flags & gen.FLAG_WIDE_SLICES else 12
        assert r["slices"] == 12 * per_picture and r["macroblocks"] > 0 and r["entries"] > 0
        assert r["rejected"] == 0 and r["unseen"] == 0
        if not flags & gen.FLAG_ODD_HEADERS:  # (user data / extension payloads hold marker values the reference acts on)
            assert r["phantom"] == 0


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_embedded_clips(clip, harness):
    """The reference's two embedded clips, through the transport-stream demultiplexer of the oracle."""
    r = run(harness, os.path.join(ROOT, "tests", "golden", clip + ".ts"), "ts")
    assert r["slices"] > 100 and r["rejected"] == 0 and r["unseen"] == 0 and r["phantom"] == 0


def test_damaged_streams(harness, tmp_path):
    """Bit flips inside slices: the parser stops where the reference's does (run past the picture, an invalid code, a block
    that overruns) with the same macroblocks before that.  The harness pairs a slice with the oracle's by the stream position
    of its start code: `unseen` are start codes the reference ran through inside a damaged slice, `phantom` slices it saw
    where there is no byte-aligned start code -- neither is compared, both must stay rare."""
    b = gen.Batch(0, 16, 6, 12, 0)
    rng = np.random.default_rng(7)
    total = {"slices": 0, "unseen": 0, "phantom": 0, "rejected": 0}
    for k in range(16):
        es = b.es(k).copy()
        done = 0
        while done < 2:
            at = int(rng.integers(200, es.size - 200))
            if es[at - 3:at + 4].min() == 0:  # keep start codes whole
                continue
            es[at] ^= 1 << int(rng.integers(0, 8))
            done += 1
        f = tmp_path / f"d{k}.es"
        es.tofile(f)
        r = run(harness, str(f))
        for key in total:
            total[key] += r[key]
    assert total["slices"] > 1000 and total["unseen"] <= 40, total


def test_handmade_stuffing_and_escapes(harness, tmp_path):
    """Any number of macroblock_stuffing codes (the reference loops, player.cpp:1268-1270) and of address escapes: every
    record -- address, vector -- equals the oracle's, no status bit (a count that carried into the motion code field gave
    a wrong vector from 16 codes on, unflagged)."""
    f = tmp_path / "stuffing.es"
    f.write_bytes(common.stuffing_es())
    p = subprocess.run([harness, str(f)], capture_output=True, text=True, timeout=300, env=dict(os.environ, EFX_HARNESS_CLEAN="1"))
    assert p.returncode == 0 and p.stdout.startswith("OK slices=37 macroblocks=1056 "), p.stdout + p.stderr


def test_selftest_region_overrun_stuffing_escapes(harness):
    """Hand-built slices no decodable stream reaches (tests/parse_harness.cpp: selftest): a slice whose stream words outgrow
    its region never touches the slots behind it in pass 2 (they are another slice's, maybe another stream's), stuffing
    runs up to 1000 codes, address escapes up to 70 000 (the increment stops the lane instead of wrapping)."""
    p = subprocess.run([harness, "--selftest"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "SELFTEST OK" in p.stdout, p.stdout + p.stderr
