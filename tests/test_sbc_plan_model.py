"""The design of the frame-parallel SBC decoder's GENERAL path (espflix_amd/csrc/k_sbc.hip: k_sbc_frames -> k_sbc_plan ->
k_sbc_gen) as an executable model on the CPU: tools/exp/sbc_general_proto.py restates, chunk by chunk and with the kernels'
own tables (frame info, frame plan, slots, row map), what they compute -- the prefix scans that replace the reference's
frame-to-frame chain (sbc_decoder.cpp:346-373: a rejected frame is synthesised from the samples and under the geometry the
state holds), the IQUANT division as a multiplication, chunks decoded in any order.  Against the test oracle on mutated
streams of every test format, in one call and in three with the state carried over, with and without decode_audio()'s probe.
(No GPU: the kernels themselves are held against the oracle in tests/test_gpu_sbc.py.)"""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("sbc_general_proto", os.path.join(ROOT, "tools", "exp", "sbc_general_proto.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plan_scans_and_chunk_decoding_reproduce_the_oracle(model, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", ["sbc_general_proto.py", "56"])
    assert model.main() == 0
    assert "mismatches 0" in capsys.readouterr().out


def test_division_by_multiplication_is_exact_for_every_width(model):
    """IQUANT divides by 2^bits - 1 (sbc_decoder.cpp:263-270); the kernels multiply by the round-up constant of Granlund &
    Montgomery.  The model asserts q == a // d on every call; here the corners of every width, the wrapped-negative dividend
    (bits 16, scale 15) included."""
    for bits in range(1, 17):
        for v in {0, 1, 2, (1 << bits) - 1, (1 << bits) - 2, 1 << (bits - 1), (1 << (bits - 1)) - 1} & set(range(1 << bits)):
            for scale in (0, 1, 7, 14, 15):
                model.iquant(v, bits, scale)
