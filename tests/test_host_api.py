"""CPU-side checks: the drop-in package surface, the embedder restatement against frames captured
from the reference, and that the C-ABI library exports what include/echoseal_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree():
    import echoseal_amd._native as nat
    hdr = open(os.path.join(ROOT, "include", "echoseal_hip.h")).read()
    declared = set(re.findall(r"\b(es_[a-z0-9_]+)\s*\(", hdr)) - {"es_ctx"}
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert os.path.exists(nat.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(nat.LIB_PATH)                       # loads without a GPU; no compute calls here
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.es_abi_version() == nat.ES_ABI_VERSION == int(re.search(r"#define\s+ES_ABI_VERSION\s+(\d+)", hdr).group(1)) == 2
    for const, val in (("ES_FRAME_LEN", 1215), ("ES_MAX_TAPS", 576), ("ES_MAX_TAPS_FAST", 160), ("ES_MAX_PEAKS", 32), ("ES_PN_BYTES", 152)):
        assert int(re.search(rf"#define\s+{const}\s+(\d+)", hdr).group(1)) == val == getattr(nat, const)


def test_loader_refuses_another_abi_version(tmp_path, monkeypatch):
    """A library built from an older header (ABI 1: 160-float tap rows) must not be bound: _native.load() raises instead."""
    import subprocess
    import echoseal_amd._native as nat
    src = tmp_path / "stale.c"
    src.write_text("int es_abi_version(void) { return 1; }\n")
    so = tmp_path / "libstale.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(nat, "LIB_PATH", str(so))
    monkeypatch.setattr(nat, "_lib", None)
    with pytest.raises(nat.NativeError, match="ABI version 1"):
        nat.load()


def test_product_never_imports_oracle():
    """Only tests/ (the sweeps under tests/fuzz/ included), __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/:
    neither the package nor the measurement tools under tools/ do."""
    for top in ("echoseal_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_dropin_surface():
    import rtwm
    from rtwm.detector import (WatermarkDetector, mseq_63, FRAME_LEN, PRE_L, HDR_L, PRE_BITS, HDR_BITS, HDR_REPEAT,
                               TIGHT_DELTA, WIDE_DELTA, EPS, BAND_PLAN, choose_band, butter_bandpass, resample_to, N_DEFAULT)
    from rtwm.embedder import WatermarkEmbedder, TxParams
    from rtwm.polar_fast import N_DEFAULT as N2, K_DEFAULT, encode, decode
    from rtwm.utils import lin_to_db, db_to_lin
    from rtwm.fastpolar import PolarCode
    from rtwm.reliability_polar_bits import Q_Nmax
    assert (FRAME_LEN, PRE_L, HDR_L, HDR_BITS, HDR_REPEAT, TIGHT_DELTA, WIDE_DELTA) == (1215, 63, 128, 16, 8, 3, 200)
    assert N_DEFAULT == N2 == 1024 and K_DEFAULT == 448 and len(Q_Nmax.split()) == 1024
    assert rtwm.WatermarkDetector is WatermarkDetector
    with pytest.raises(ValueError):
        WatermarkDetector(b"short")
    for bad in (dict(N=1000, K=448), dict(N=1024, K=0), dict(N=1024, K=448, list_size=0), dict(N=1024, K=448, crc_size=448)):
        with pytest.raises(ValueError):
            PolarCode(**bad)
    pc = PolarCode(1024, 448)
    assert pc._info_len == 440 and pc.frozen.sum() == 576 and pc._data_pos[0] == 0
    with pytest.raises(ValueError):
        pc.encode(np.zeros(100, np.uint8))
    with pytest.raises(ValueError):
        encode(b"x" * 54)
    assert abs(db_to_lin(lin_to_db(0.5)) - 0.5) < 1e-9


def test_embedder_reproduces_reference_frames(golden_detector):
    """Clean frames in the fixture were produced by the reference's embedder with a fixed payload."""
    from echoseal_amd.embedder import WatermarkEmbedder
    g = golden_detector
    seen = 0
    for i in range(int(g["det/count"])):
        t = f"det/{i:02d}"
        key = g[f"{t}/key"].tobytes(); ctr = int(g[f"{t}/ctr"])
        mine = WatermarkEmbedder(key).make_frames([ctr], [g[f"{t}/payload"].tobytes()])[0]
        ref = g[f"{t}/x"]
        if np.array_equal(mine, ref):
            seen += 1
        else:                                             # noisy variants carry added AWGN
            assert np.std(mine - ref) > 0.1
    assert seen >= 14


def test_embedder_alignment_like_reference_test():
    """Mirror of the reference's tests/test_embedder_detector_alignment.py:22-33."""
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.detector import WatermarkDetector, FRAME_LEN, PRE_L, HDR_L
    key = bytes(32)
    tx = WatermarkEmbedder(key)
    try:
        rx = WatermarkDetector(key)
    except Exception as e:                                # needs the GPU engine only lazily
        pytest.fail(f"constructing the detector must not need a GPU: {e}")
    np.testing.assert_allclose(tx._preamble_sy, rx._pre_sy)
    np.testing.assert_allclose(tx._hdr_pn_sy, rx._hdr_pn_sy)
    for ctr in (0, 1, 255, 1024):
        assert np.array_equal(tx.sec.pn_bits(ctr, PRE_L + HDR_L + tx.p.N), rx.sec.pn_bits(ctr, FRAME_LEN))
    assert tx.process(np.zeros(3000, np.float32)).shape == (3000,)


def test_hot_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rtwm.polar_fast import decode
    from echoseal_amd._native import NativeError
    with pytest.raises(NativeError):
        decode(np.ones(1024))


def test_declared_limits_raise_not_implemented():
    """The documented limits of the drop-in are an explicit error contract, raised BEFORE any GPU work:
    the HIP decoder serves Polar(1024, K)+CRC-8 for every K (the reference's detector instantiates 448 only, rtwm/polar_fast.py:18-24)
    and the detector any fs_target whose matched filters fit 576 taps (44 100 Hz and up; rtwm/detector.py:27).  Constructing such objects works,
    as in the reference."""
    from rtwm.fastpolar import PolarCode
    from rtwm.detector import WatermarkDetector
    with pytest.raises(ValueError):                                            # as in the reference: the reliability table has 1024 entries
        PolarCode(512, 224)
    for n, k, crc in ((1024, 300, 8), (1024, 12, 8), (1024, 1024, 8)):     # any K: served (GPU tests); the TX side is generic host code
        pc = PolarCode(n, k, crc_size=crc)
        assert pc.encode(np.zeros(k - crc, np.uint8)).shape == (n,)
    with pytest.raises(NotImplementedError, match="CRC-8"):                    # (the reference's own encode / decode only work with crc_size 8)
        PolarCode(1024, 448, crc_size=16).decode(np.ones(1024))
    det = WatermarkDetector(bytes(32), fs_target=44_100)                       # any rate the band plan admits is served (GPU tests): 550 taps here
    assert det.fs_target == 44_100
    import torch
    if torch.cuda.is_available():
        with pytest.raises(NotImplementedError, match="taps"):                 # the 18-22 kHz band within 50 Hz of Nyquist: more than 576 taps
            WatermarkDetector(bytes(32), fs_target=44_050).verify(np.zeros(5000, np.float32), 44_050)
    # list sizes: any value >= 1 is accepted, as in the reference; the HIP kernels serve up to 256
    assert PolarCode(1024, 448, list_size=3).list_size == 3 and PolarCode(1024, 448, list_size=1000).list_size == 1000


def test_wav_ingest_round_trip(tmp_path):
    """f-4 ingest: a PCM16 WAV comes back as the int16 samples that were written, and x / 32768 in float32 is exactly the
    value soundfile.read hands the reference for such a file."""
    from echoseal_amd.audiofile import read_wav, write_wav_pcm16
    rng = np.random.default_rng(1)
    x = rng.integers(-32768, 32768, 5000).astype(np.int16)
    p = str(tmp_path / "a.wav")
    write_wav_pcm16(p, x, 48_000)
    y, fs = read_wav(p)
    assert fs == 48_000 and y.dtype == np.int16 and np.array_equal(x, y)
    f = rng.uniform(-1, 1, 3000)
    write_wav_pcm16(p, f, 44_100)
    y, fs = read_wav(p)
    assert fs == 44_100 and np.max(np.abs(y.astype(np.float64) / 32768.0 - f)) <= 0.5 / 32768 + 1e-12
    assert np.array_equal((y.astype(np.float64) / 32768.0).astype(np.float32), y.astype(np.float32) / np.float32(32768.0))


def test_undecoded_records_raise_not_pass_silently():
    """ncand < 0 (a list-decoder block that found no scratch-slab slot) must surface as an exception wherever results are consumed."""
    import torch
    from echoseal_amd._native import NativeError
    from echoseal_amd.engine import SclResult, select_payload
    z = torch.zeros
    res = SclResult(z((3, 55), dtype=torch.uint8), z(3, dtype=torch.uint8), z((3, 8, 55), dtype=torch.uint8), z((3, 8), dtype=torch.float64),
                    z((3, 8), dtype=torch.uint8), torch.tensor([8, -1, 0], dtype=torch.int32))
    with pytest.raises(NativeError, match="not decoded"):
        res.check()
    with pytest.raises(NativeError, match="not decoded"):
        select_payload(res, 1)
    res.ncand[1] = 8
    assert res.check() is res


def test_tools_scripts_at_least_parse():
    """tools/ holds measurement scripts that only run on the GPU box; here: every Python one compiles, every shell one passes `bash -n`, and
    none refers to a file of this repository that does not exist (the round-3 tree kept tools for deleted experiments)."""
    import py_compile
    import subprocess
    tools = os.path.join(ROOT, "tools")
    for dirpath, _, files in os.walk(tools):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                py_compile.compile(path, doraise=True)
            elif f.endswith(".sh"):
                assert subprocess.run(["bash", "-n", path]).returncode == 0, path
            if f.endswith((".py", ".sh")):
                for ref in re.findall(r"tools/[\w/]+\.(?:py|sh|hip)", open(path).read()):
                    assert os.path.exists(os.path.join(ROOT, ref)), f"{path} refers to {ref}"
