"""HIP kernels (through the C ABI) against the round-2 REFERENCE fixtures (tests/golden/*, captured by running the
reference: oracle/refshim/gen_golden_r2.py).  Bars: sync offsets / decoded bits exact, thr <= 1e-12, LLR <= 1e-5."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from echoseal_amd.crypto import SecureChannel
from echoseal_amd.tables import pack_tables

HERE = os.path.dirname(os.path.abspath(__file__))
KEY = b"\xAA" * 32


def _g(name):
    return np.load(os.path.join(HERE, "golden", name))


def _dev(eng, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(eng.device) for a in arrs]


def test_c3_windows_sync_header_llr_vs_reference(engine):
    g = _g("c3_windows.npz")
    n = g["win"].shape[0]
    win, band = _dev(engine, g["win"], g["band"])
    sec = SecureChannel(KEY)
    pn, = _dev(engine, sec.pn_bytes_batch([int(c) for c in g["ctr"]], 152))
    hdr_pn, = _dev(engine, np.packbits(sec.pn_bits(0, 128)).reshape(1, 16))
    for name in ("fast", "f64"):
        sy = engine.sync_fast(win, band) if name == "fast" else engine.sync(win, band, keep_corr=True)
        thr = sy.thr.cpu().numpy(); pk = sy.peaks.cpu().numpy(); npk = sy.npeaks.cpu().numpy()
        assert np.max(np.abs(thr - g["thr"])) < 1e-12, name
        for i in range(n):
            k = min(int(g["npeaks"][i]), 32)
            assert (int(npk[i]) & 0xFFFF) == k and bool(npk[i] >> 30) == bool(g["fallback"][i]), (name, i)
            assert list(pk[i, :k]) == list(g["peaks"][i, :k]), (name, i)          # sync offsets: exact
        if name == "f64":
            corr = sy.corr.cpu().numpy()
            for i in range(16):
                assert np.max(np.abs(corr[i] - g[f"corr/{i:03d}"])) < 1e-12
    y = sy.y
    # header decode at the peaks the reference visited
    rows, starts, want = [], [], []
    for i in range(n):
        vis = [int(p) for p in g["peaks"][i, :min(int(g["npeaks"][i]), 25)] if p + 1215 <= 2048][:5]
        for j, st in enumerate(vis):
            rows.append(i); starts.append(st); want.append(g["hdr"][i, j])
    rows_t = torch.tensor(rows, device=engine.device)
    ok, val, score = engine.header(y[rows_t].contiguous(), band[rows_t].contiguous(), hdr_pn,
                                   start=torch.tensor(starts, dtype=torch.int32, device=engine.device))
    want = np.array(want)
    assert np.array_equal(ok.cpu().numpy().astype(bool), want[:, 0].astype(bool))
    assert np.array_equal(val.cpu().numpy(), want[:, 1].astype(np.int64))
    assert np.max(np.abs(score.cpu().numpy() - want[:, 2]) / np.maximum(1.0, np.abs(want[:, 2]))) <= 1e-4
    # _llr at the reference's first peak (frame may be cut short by the window end)
    start = torch.from_numpy(g["peaks"][:, 0].astype(np.int32)).to(engine.device)
    worst = amb = 0
    for variant, key in ((0, "llr0"), (1, "llr1")):
        llr, best_s, score2 = engine.llr(y, band, pn, start=start, variant=variant, want_diag=True)
        llr = llr.cpu().numpy(); best_s = best_s.cpu().numpy(); sc = score2.cpu().numpy()
        for i in range(n):
            if 2048 - int(g["peaks"][i, 0]) <= 191:
                assert not llr[i].any() and not g[key][i].any()
                continue
            if best_s[i] != g["best_s"][i, variant]:
                assert (sc[i, 0] - sc[i, 1]) / max(abs(sc[i, 0]), 1e-30) < 1e-5, (i, variant)     # tie-ambiguous
                amb += 1
                continue
            worst = max(worst, float(np.max(np.abs(llr[i] - g[key][i]))))
    assert worst <= 1e-5 and amb == 0, (worst, amb)          # observed: no shift differs (tests/golden/h1_margin_report.json)


@pytest.mark.parametrize("fs", [44_100, 96_000])
def test_other_fs_target_vs_reference(oracle, fs):
    """An engine whose tables are built for fs_target = 44 100 / 96 000 (matched filters of up to 550 taps: the demodulator's and the header
    decoder's long-filter instantiations) against the reference built for that rate, and against the oracle bit for bit; then the drop-in
    detector at that rate runs verify() end to end."""
    from echoseal_amd.engine import RxEngine
    from echoseal_amd.tables import pack_tables
    g = _g(f"fs{fs}_windows.npz")
    eng = RxEngine(0, list_size_max=8, fs=fs)
    ba, tpl, taps, ntaps, _ = pack_tables(fs)
    n = g["win"].shape[0]
    win, band = _dev(eng, g["win"], g["band"])
    sec = SecureChannel(KEY)
    pn, = _dev(eng, sec.pn_bytes_batch([int(c) for c in g["ctr"]], 152))
    hdr_pn, = _dev(eng, np.packbits(sec.pn_bits(0, 128)).reshape(1, 16))
    sy = eng.sync_fast(win, band)
    thr = sy.thr.cpu().numpy(); pk = sy.peaks.cpu().numpy(); npk = sy.npeaks.cpu().numpy()
    assert np.max(np.abs(thr - g["thr"])) < 1e-12
    for i in range(n):
        k = min(int(g["npeaks"][i]), 32)
        assert (int(npk[i]) & 0xFFFF) == k and bool(npk[i] >> 30) == bool(g["fallback"][i]), i
        assert list(pk[i, :k]) == list(g["peaks"][i, :k]), i
    y = sy.y
    rows, starts, want = [], [], []
    for i in range(n):
        vis = [int(p) for p in g["peaks"][i, :min(int(g["npeaks"][i]), 25)] if p + 1215 <= 2048][:5]
        for j, st in enumerate(vis):
            rows.append(i); starts.append(st); want.append(g["hdr"][i, j])
    rows_t = torch.tensor(rows, device=eng.device)
    ok, val, score = eng.header(y[rows_t].contiguous(), band[rows_t].contiguous(), hdr_pn,
                                start=torch.tensor(starts, dtype=torch.int32, device=eng.device))
    want = np.array(want)
    assert np.array_equal(ok.cpu().numpy().astype(bool), want[:, 0].astype(bool))
    assert np.array_equal(val.cpu().numpy(), want[:, 1].astype(np.int64))
    assert np.max(np.abs(score.cpu().numpy() - want[:, 2]) / np.maximum(1.0, np.abs(want[:, 2]))) <= 1e-4
    start = torch.from_numpy(g["peaks"][:, 0].astype(np.int32)).to(eng.device)
    yh = y.cpu().numpy()
    worst = 0.0
    for variant, key in ((0, "llr0"), (1, "llr1")):
        llr, best_s, score2 = eng.llr(y, band, pn, start=start, variant=variant, want_diag=True)
        llr = llr.cpu().numpy(); best_s = best_s.cpu().numpy(); sc = score2.cpu().numpy()
        for i in range(n):
            st = int(g["peaks"][i, 0]); b = int(g["band"][i])
            pnb = sec.pn_bits(int(g["ctr"][i]), 1215)
            o_llr, o_s, _, _ = oracle.llr(yh[i, st:st + 1215], pnb[191:1215] if variant == 0 else pnb[:1024], taps[b, :ntaps[b]])
            assert np.array_equal(o_llr, llr[i]) and (2048 - st <= 191 or o_s == int(best_s[i])), (i, variant)      # HIP == oracle, bit for bit
            if 2048 - st <= 191:
                assert not llr[i].any() and not g[key][i].any()
                continue
            if best_s[i] != g["best_s"][i, variant]:
                assert (sc[i, 0] - sc[i, 1]) / max(abs(sc[i, 0]), 1e-30) < 1e-5, (i, variant)
                continue
            worst = max(worst, float(np.max(np.abs(llr[i] - g[key][i]))))
    assert worst <= 1e-5, worst
    eng.close()
    from rtwm.detector import WatermarkDetector
    det = WatermarkDetector(KEY, fs_target=fs, list_size=8)
    assert det.verify(g["win"][:3].reshape(-1).astype(np.float32), fs) in (True, False) and det.engine.fs == fs


def test_multi_peak_records_vs_reference(engine):
    g = _g("sync_multi.npz")
    for i in range(int(g["count"])):
        t = f"{i:02d}"
        x, band = _dev(engine, g[f"{t}/x"].reshape(1, -1), np.array([int(g[f"{t}/band"])], np.uint8))
        ref = list(g[f"{t}/peaks"])
        for name in ("fast", "f64"):
            sy = engine.sync_fast(x, band) if (name == "fast" and x.shape[1] - 62 <= engine.FAST_MAX_LAGS) else engine.sync(x, band)
            assert abs(float(sy.thr[0]) - float(g[f"{t}/thr"])) < 1e-12
            k = int(sy.npeaks[0]) & 0xFFFF
            assert k == min(len(ref), 32) and bool(int(sy.npeaks[0]) >> 30) == bool(g[f"{t}/fallback"])
            assert list(sy.peaks[0, :k].cpu().numpy()) == ref[:k], (t, name)


@pytest.mark.parametrize("multi", [0, 1, 2, 3])
def test_polar_bulk_vs_reference(engine, multi):
    """1 024 LLR vectors: final SCL-8 lists bit-identical (bits, metrics, CRC flags) to the reference run on the C
    library's exp/log1p; hard-decision shortcut and (info, ok) through es_select_batch.  All four mappings: one frame
    per wave (0), 16 paths x 4 lanes per wave (1), 32 paths x 2 lanes per wave (2), 64 paths x 1 lane per wave (3: es_scl_wide.hip,
    the kernel of the grouped headline and of the large launches)."""
    g = _g("polar_bulk_glibc.npz")
    llr, = _dev(engine, g["llr"])
    engine.set_option("scl_multi", 1 if multi else 0)
    engine.set_option("scl_lanes", {0: 4, 1: 4, 2: 2, 3: 1}[multi])
    try:
        res = engine.scl(llr, list_size=8, skip_if_hard_ok=False)
        short = engine.scl(llr, list_size=8, skip_if_hard_ok=True)
    finally:
        engine.set_option("scl_multi", -1); engine.set_option("scl_lanes", 0)
    took = g["took_list"]
    assert np.array_equal(short.ncand.cpu().numpy() > 0, took)                   # the shortcut fires exactly where the reference's did
    assert np.array_equal(res.cand_info.cpu().numpy()[took], g["cand_info"][took])
    assert np.array_equal(res.cand_metric.cpu().numpy()[took].view(np.uint64), g["cand_metric"][took].view(np.uint64))
    assert np.array_equal(res.cand_ok.cpu().numpy()[took], g["cand_crc"][took])
    payload, ok, which = engine.select(short)
    assert np.array_equal(payload.cpu().numpy(), g["info"])
    assert np.array_equal(ok.cpu().numpy() == 1, g["ok"])


@pytest.mark.parametrize("L", [1, 4, 16])
def test_polar_sweep_vs_reference(engine, L):
    """BASELINE config 5's list sizes (1 / 4 / 16; 8 is the bulk test above): 256 vectors per size, final lists bit-identical (bits, metrics,
    CRC flags) to the reference run on the C library's exp/log1p, on every mapping that serves the size; (info, ok) through es_select_batch."""
    g = _g("polar_sweep_glibc.npz")
    llr, = _dev(engine, g[f"L{L}/llr"])
    took = g[f"L{L}/took_list"]
    for multi in (0, 1, 2, 3):
        engine.set_option("scl_multi", 1 if multi else 0)
        engine.set_option("scl_lanes", {0: 4, 1: 4, 2: 2, 3: 1}[multi])
        try:
            res = engine.scl(llr, list_size=L, skip_if_hard_ok=False)
            short = engine.scl(llr, list_size=L, skip_if_hard_ok=True)
        finally:
            engine.set_option("scl_multi", -1); engine.set_option("scl_lanes", 0)
        assert np.array_equal(short.ncand.cpu().numpy() > 0, took), multi
        assert np.array_equal(res.cand_info.cpu().numpy()[took], g[f"L{L}/cand_info"][took]), multi
        assert np.array_equal(res.cand_metric.cpu().numpy()[took].view(np.uint64), g[f"L{L}/cand_metric"][took].view(np.uint64)), multi
        assert np.array_equal(res.cand_ok.cpu().numpy()[took], g[f"L{L}/cand_crc"][took]), multi
        payload, ok, which = engine.select(short)
        assert np.array_equal(payload.cpu().numpy(), g[f"L{L}/info"]), multi
        assert np.array_equal(ok.cpu().numpy() == 1, g[f"L{L}/ok"]), multi


@pytest.mark.parametrize("name", ["polar_codes", "polar_codes2"])
def test_other_codes_vs_reference(name):
    """PolarCode(1024, K) for K = 16, 64, 200, 512, 1000 and 9, 13, 301, 1023, 1024 (the reference's class takes any K; its detector uses 448): the run-time-K
    instantiation of the lane-per-path kernel against the reference's lists (C-library exp/log1p): bits, metrics and CRC flags identical, and
    (info, ok) through the host-side tail of PolarCode.decode."""
    from echoseal_amd.engine import RxEngine, select_payload
    g = _g(f"{name}_glibc.npz")
    for K in g["ks"]:
        eng = RxEngine(0, list_size_max=64, code_k=int(K))
        assert eng.info_bytes == (int(K) - 8 + 7) // 8
        llr, = _dev(eng, g[f"K{K}/llr"])
        for L in g["lists"]:
            t = f"K{K}/L{L}"
            took = g[f"{t}/took_list"]
            res = eng.scl(llr, list_size=int(L), skip_if_hard_ok=False).check()
            short = eng.scl(llr, list_size=int(L), skip_if_hard_ok=True).check()
            assert np.array_equal(short.ncand.cpu().numpy() > 0, took), t
            nc = g[f"{t}/ncand"]
            for i in np.flatnonzero(took):
                n = int(nc[i])
                assert int(res.ncand[i]) == n, (t, i)
                assert np.array_equal(res.cand_info[i, :n].cpu().numpy(), g[f"{t}/cand_info"][i, :n]), (t, i)
                assert np.array_equal(res.cand_metric[i, :n].cpu().numpy().view(np.uint64), g[f"{t}/cand_metric"][i, :n].view(np.uint64)), (t, i)
                assert np.array_equal(res.cand_ok[i, :n].cpu().numpy(), g[f"{t}/cand_crc"][i, :n]), (t, i)
            for i in range(llr.shape[0]):
                payload, ok = select_payload(short, i, None)
                assert ok == bool(g[f"{t}/ok"][i]) and payload == g[f"{t}/info"][i].tobytes(), (t, i)
        eng.close()


def test_decode_with_validator_vs_reference(engine):
    from echoseal_amd.engine import select_payload
    from test_golden_r2 import _validator
    g = _g("polar_validator.npz")
    n = int(g["count"])
    key = SecureChannel(KEY)._aead._key
    by_L = {}
    for i in range(n):
        by_L.setdefault(int(g[f"{i:03d}/L"]), []).append(i)
    for L, idx in by_L.items():
        llr, = _dev(engine, np.stack([g[f"{i:03d}/llr"] for i in idx]))
        res = engine.scl(llr, list_size=L, skip_if_hard_ok=False)
        for r, i in enumerate(idx):
            t = f"{i:03d}"
            spec = str(g[f"{t}/spec"]); arg = g[f"{t}/arg"]
            seen = []
            payload, ok = select_payload(res, r, _validator(spec, arg, KEY, seen))
            assert ok == bool(g[f"{t}/ok"]) and payload == g[f"{t}/info"].tobytes(), (i, spec)
            assert seen == [s.tobytes() for s in g[f"{t}/seen"]], (i, spec)
        # the detector's own validator on the GPU (es_select_batch with the AEAD key)
        aead = [(r, i) for r, i in enumerate(idx) if str(g[f"{i:03d}/spec"]) == "aead"]
        if aead:
            want_ctr = np.zeros(len(idx), np.int64)
            for r, i in aead:
                want_ctr[r] = int(g[f"{i:03d}/arg"])
            payload, ok, which = engine.select(res, key32=key, ctrs=torch.from_numpy(want_ctr))
            payload = payload.cpu().numpy(); ok = ok.cpu().numpy()
            for r, i in aead:
                assert (ok[r] == 1) == bool(g[f"{i:03d}/ok"]) and payload[r].tobytes() == g[f"{i:03d}/info"].tobytes(), i


def test_verify_3s_clip_follows_reference(engine):
    from echoseal_amd.detector import WatermarkDetector
    g = _g("verify3s.npz")
    det = WatermarkDetector(KEY, list_size=1, engine=engine)
    det._trace = []
    det._hdr_trace = []
    assert det.verify(g["clip"], 48_000) == bool(g["result"])
    assert np.array_equal(np.array(det._trace, dtype=np.int64).reshape(-1, 3), g["trace"])
    hdr = np.array(det._hdr_trace, dtype=np.float64).reshape(-1, 3)
    assert hdr.shape == g["hdr"].shape
    assert np.array_equal(hdr[:, :2], g["hdr"][:, :2])
    assert np.max(np.abs(hdr[:, 2] - g["hdr"][:, 2]) / np.maximum(1.0, np.abs(g["hdr"][:, 2]))) <= 1e-4


def test_list_sizes_that_are_not_powers_of_two(engine, oracle):
    """Any list size 1..256 (the reference takes any): the next power of two's kernel with the surplus paths switched
    off.  Against the reference fixtures (3, 5, 6, 12, 24, 100) and, for more sizes and frames, against the oracle."""
    g = _g("polar_odd_glibc.npz")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/llr")})
    big = RxEngineBig.get(engine)
    llr = np.stack([np.asarray(g[f"{n}/llr"], np.float64) for n in names])
    for L in (3, 5, 6, 12, 24, 100):
        res = big.scl(torch.from_numpy(llr).to(big.device), list_size=L, skip_if_hard_ok=False)
        for r, n in enumerate(names):
            if f"{n}/L{L}/cand_metric" in g.files:
                assert np.array_equal(res.cand_info[r].cpu().numpy(), g[f"{n}/L{L}/cand_info"]), (n, L)
                assert np.array_equal(res.cand_metric[r].cpu().numpy().view(np.uint64), g[f"{n}/L{L}/cand_metric"].view(np.uint64)), (n, L)
                assert np.array_equal(res.cand_ok[r].cpu().numpy(), g[f"{n}/L{L}/cand_crc"])
    rng = np.random.default_rng(77)
    x = np.clip(rng.normal(0, 3, (40, 1024)), -12, 12).astype(np.float32)
    x[0] = 0.0; x[0, 0] = 1e-3
    for L in (1, 2, 3, 4, 7, 9, 15, 16, 17, 31, 33, 48, 65, 127, 129, 200, 255):
        for multi in ((0, 1, 2) if L <= 16 else (0,)):
            big.set_option("scl_multi", 1 if multi else 0); big.set_option("scl_lanes", 2 if multi == 2 else 4)
            res = big.scl(torch.from_numpy(x).to(big.device), list_size=L, skip_if_hard_ok=False)
            big.set_option("scl_multi", -1); big.set_option("scl_lanes", 0)
            assert res.cand_metric.shape == (40, L)
            for i in range(0, 40, 3 if L < 64 else 13):
                nn, ci, cm, cc = oracle.scl_list(x[i].astype(np.float64), L)
                assert int(res.ncand[i]) == L
                assert np.array_equal(np.packbits(ci, axis=1), res.cand_info[i].cpu().numpy()), (L, multi, i)
                assert np.array_equal(cm, res.cand_metric[i].cpu().numpy()) and np.array_equal(cc, res.cand_ok[i].cpu().numpy())


class RxEngineBig:
    """An engine created for list sizes up to 256 (shared by the tests of this module)."""
    _eng = None

    @classmethod
    def get(cls, engine):
        if cls._eng is None:
            from echoseal_amd.engine import RxEngine
            cls._eng = RxEngine(engine.device.index, list_size_max=256)
        return cls._eng
