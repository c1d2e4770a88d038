"""Pin the C oracle (oracle/c/eso_polar.c) against the reference's own outputs.

tests/golden/polar_{default,glibc}.npz were produced by RUNNING rtwm/fastpolar.py (generator:
oracle/refshim/gen_golden.py): `default` = NumPy as installed on the build host (AVX-512 exp/log1p
in the penalty), `glibc` = NumPy forced onto the C library.  Decoded bits must agree in both; the
path metrics agree bit-for-bit in glibc mode and to a few ulp in default mode.
"""
import hashlib
import os

import numpy as np
import pytest

LISTS = (1, 4, 8, 16, 32)


def _cases(g):
    return sorted({k.split("/")[0] for k in g.files if k.endswith("/llr")})


def test_frozen_set_and_encode_known_answers(oracle):
    frozen, dpos = oracle.polar_tables()
    assert hashlib.sha256(frozen.tobytes()).hexdigest().startswith("e41c7b11e70f3aa7b0a5501c416c10ac")
    assert list(dpos[:8]) == list(range(8)) and list(dpos[-8:]) == [774, 776, 777, 778, 784, 800, 832, 896]
    assert not frozen[0] and frozen[1023]                       # reference quirk
    bits = np.unpackbits(np.frombuffer(bytes(range(55)), np.uint8))
    assert format(oracle.crc8(bits), "08b") == "01000000"
    code = oracle.polar_encode(bits)
    assert hashlib.sha256(code.tobytes()).hexdigest().startswith("1c6b1d9e")


def test_decode_matches_reference(oracle, golden_polar):
    mode, g = golden_polar
    for name in _cases(g):
        llr = g[f"{name}/llr"]
        for L in LISTS:
            info, ok, took = oracle.polar_decode(llr, L)
            assert ok == bool(g[f"{name}/L{L}/ok"]), (mode, name, L)
            assert np.array_equal(np.packbits(info), g[f"{name}/L{L}/info"]), (mode, name, L)
            assert took == (f"{name}/L{L}/cand_metric" in g.files), (mode, name, L)


def test_candidate_lists_match_reference(oracle, golden_polar):
    mode, g = golden_polar
    checked = 0
    for name in _cases(g):
        llr = g[f"{name}/llr"]
        for L in LISTS:
            key = f"{name}/L{L}/cand_metric"
            if key not in g.files:
                continue
            n, ci, cm, cc = oracle.scl_list(llr, L)
            ref_m = g[key]
            assert n == ref_m.size
            if mode == "glibc":
                assert np.array_equal(cm.view(np.uint64), ref_m.view(np.uint64)), (name, L)
            else:
                assert np.allclose(cm, ref_m, rtol=1e-13, atol=0), (name, L)
            assert np.array_equal(np.packbits(ci, axis=1), g[f"{name}/L{L}/cand_info"]), (mode, name, L)
            assert np.array_equal(cc, g[f"{name}/L{L}/cand_crc"]), (mode, name, L)
            checked += 1
    assert checked >= 40


def test_reference_test_vectors(oracle, golden_polar):
    """The reference's own seeded polar tests (tests/test_polar.py:64-109): AWGN sigma 0.15 round
    trips through the hard-decision shortcut; +-10 / +-2 clean codewords decode to the payload."""
    _, g = golden_polar
    for seed in (1234, 4321):
        rng = np.random.default_rng(seed)
        if seed == 1234:
            payload = np.packbits(rng.integers(0, 2, 440, dtype=np.uint8)).tobytes()
        else:
            payload = rng.integers(0, 256, size=55, dtype=np.uint8).tobytes()
        info, ok, took = oracle.polar_decode(g[f"awgn015_seed{seed}/llr"], 8)
        assert ok and not took and np.packbits(info).tobytes() == payload
    info, ok, _ = oracle.polar_decode(g["clean10_range55/llr"], 8)
    assert ok and np.packbits(info).tobytes() == bytes(range(55))
    info, ok, _ = oracle.polar_decode(g["clean2_A55/llr"], 8)
    assert ok and np.packbits(info).tobytes() == b"A" * 55
    # all-zero LLR: hard decision is the all-zero word, whose CRC is 0 -> accepted by the shortcut
    info, ok, took = oracle.polar_decode(np.zeros(1024), 8)
    assert ok and not took and not info.any()


@pytest.mark.parametrize("L", [1, 2, 4, 8])
def test_all_candidates_tie(oracle, L):
    """LLR = +-0 everywhere except one position makes every candidate metric tie; the stable sort
    order (path index, bit 0 before bit 1) then decides.  Property: list is sorted, deterministic."""
    llr = np.zeros(1024); llr[0] = 1e-3                      # defeat the hard-decision shortcut
    n, ci, cm, cc = oracle.scl_list(llr, L)
    assert n == L and np.all(np.diff(cm) >= 0)
    n2, ci2, cm2, _ = oracle.scl_list(llr, L)
    assert np.array_equal(ci, ci2) and np.array_equal(cm, cm2)


@pytest.mark.parametrize("mode", ["default", "glibc"])
def test_wide_lists_match_reference(oracle, mode):
    """L = 64 and 256 (detector default) against tests/golden/polar_wide_*.npz."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"polar_wide_{mode}.npz"))
    for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/llr")}):
        for L in (64, 256):
            info, ok, took = oracle.polar_decode(g[f"{name}/llr"], L)
            assert ok == bool(g[f"{name}/L{L}/ok"]) and np.array_equal(np.packbits(info), g[f"{name}/L{L}/info"])
            key = f"{name}/L{L}/cand_metric"
            if key in g.files:
                n, ci, cm, cc = oracle.scl_list(g[f"{name}/llr"], L)
                assert np.array_equal(np.packbits(ci, axis=1), g[f"{name}/L{L}/cand_info"])
                assert np.array_equal(cm, g[key]) if mode == "glibc" else np.allclose(cm, g[key], rtol=1e-13, atol=0)


def test_quick_frame_list32_decodes_match_reference(oracle):
    """BASELINE config 1 at the list size the reference's own test uses (tests/test_roundtrip_quick.py:14): the 20 LLR vectors the
    reference's verify_raw_frame handed to polar decode on the quick-test frame (tests/golden/quick32.npz, generator
    oracle/refshim/gen_golden_quick32.py), decoded by PolarCode.decode(list_size=32) without a validator: (info, ok) bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quick32.npz"))
    assert int(g["list_size"]) == 32 and g["llr"].shape == (20, 1024) and (g["dec_list_size"] == 32).all()
    for k in range(g["llr"].shape[0]):
        info, ok, _took = oracle.polar_decode(g["llr"][k], 32)
        assert ok == bool(g["plain_ok"][k]) and np.array_equal(np.packbits(info), g["plain_info"][k]), k
