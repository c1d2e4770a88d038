"""Pin the C oracle (and the host selection logic) against the round-2 reference fixtures: outputs of the REFERENCE
itself, captured by oracle/refshim/gen_golden_r2.py (which imports and runs /root/reference in the build container).

    c3_windows.npz       256 BASELINE-config-3 windows through the reference's own _scan_band_multi_frame, _decode_header, _llr
    sync_multi.npz       records with 2-5 peaks (3-frame clips, frame-sized records with a second peak), same capture
    polar_bulk_*.npz     1 024 LLR vectors through PolarCode.decode(list_size=8), final lists, both NumPy run-time modes
    polar_sweep_*.npz    (round 3) 3 x 256 LLR vectors through PolarCode.decode at list sizes 1, 4 and 16 (BASELINE config 5's sweep)
    polar_validator.npz  PolarCode.decode with validators: every payload the validator was shown, in order
    verify3s.npz         verify() on a 3 s noisy clip: tries, header decodes (host flow: tests/test_detector.py on the GPU)

Bars (BASELINE.json north_star): sync offsets / decoded bits exact, LLR within 1e-5.  The LLR shift search is the one
place where the reference is not bit-reproducible across machines (float32 BLAS rounding): a frame whose
best / runner-up scores differ by less than 1e-5 relative is classified TIE-AMBIGUOUS when the chosen shift differs,
and the report (tests/golden/h1_margin_report.json, asserted by this test; ES_WRITE_H1_REPORT=1 rewrites it) counts them. Observed: none.
"""
import json
import os

import numpy as np
import pytest
import torch

from echoseal_amd.tables import band_coeffs, matched_filter_taps, pack_tables
from echoseal_amd.utils import BAND_PLAN

HERE = os.path.dirname(os.path.abspath(__file__))
KEY = b"\xAA" * 32


def _g(name):
    return np.load(os.path.join(HERE, "golden", name))


@pytest.fixture(scope="module")
def tables():
    return pack_tables()


def _sync(oracle, tables, x, band):
    ba, tpl, taps, ntaps, _ = tables
    y = oracle.lfilter(ba[band][:9], ba[band][9:], x)
    corr = oracle.ncc(y, tpl[band])
    thr, med, mad = oracle.cfar_threshold(corr)
    peaks, total, fb = oracle.pick_peaks(corr, thr)
    return y, corr, thr, med, mad, peaks, total, fb


def test_c3_windows_sync_matches_reference(oracle, tables):
    g = _g("c3_windows.npz")
    n = g["win"].shape[0]
    assert n == 256
    nfb = 0
    for i in range(n):
        band = int(g["band"][i])
        y, corr, thr, med, mad, peaks, total, fb = _sync(oracle, tables, g["win"][i], band)
        if f"corr/{i:03d}" in g.files:
            assert np.max(np.abs(corr - g[f"corr/{i:03d}"])) < 1e-12
        assert abs(thr - float(g["thr"][i])) < 1e-12 and abs(med - float(g["med"][i])) < 1e-12
        assert abs(mad - float(g["mad"][i])) < 1e-12
        assert fb == bool(g["fallback"][i]) and total == int(g["npeaks"][i])
        k = min(total, 32)
        assert list(peaks[:k]) == list(g["peaks"][i, :k]), i                   # sync offsets: exact
        nfb += fb
    assert nfb > 200            # at -15 dB the threshold saturates at 0.95: the top-5 fallback IS the C3 path


def test_c3_windows_header_and_llr_match_reference(oracle, tables):
    from echoseal_amd.crypto import SecureChannel
    g = _g("c3_windows.npz")
    ba, tpl, taps, ntaps, _ = tables
    sec = SecureChannel(KEY)
    hdr_pn = sec.pn_bits(0, 128)
    worst, amb, exact, rows = 0.0, 0, 0, []
    for i in range(g["win"].shape[0]):
        band = int(g["band"][i]); ctr = int(g["ctr"][i])
        h = taps[band, :ntaps[band]]
        y = oracle.lfilter(ba[band][:9], ba[band][9:], g["win"][i])
        # header decode at every peak the reference visited (frame fits), up to 5 kept in the fixture
        visited = [int(p) for p in g["peaks"][i, :min(int(g["npeaks"][i]), 25)] if p + 1215 <= y.size][:5]
        assert len(visited) == min(5, int(g["nvisited"][i]))
        for j, st in enumerate(visited):
            ok, val, score, _ = oracle.decode_header(y[st:st + 1215], hdr_pn, h)
            assert ok == bool(g["hdr"][i, j, 0]) and val == int(g["hdr"][i, j, 1]), (i, j)
            assert abs(score - g["hdr"][i, j, 2]) <= 1e-4 * max(1.0, abs(g["hdr"][i, j, 2]))
        st = int(g["peaks"][i, 0])
        pn = sec.pn_bits(ctr, 1215)
        for variant, key, pnb in ((0, "llr0", pn[191:1215]), (1, "llr1", pn[:1024])):
            llr, best_s, s0, s1 = oracle.llr(y[st:st + 1215], pnb, h)
            ref_s = int(g["best_s"][i, variant])
            margin = (s0 - s1) / max(abs(s0), 1e-30)
            if y.size - st <= 191:                                             # too short: zeros, no shift search
                assert not llr.any() and not g[key][i].any()
                continue
            rows.append(margin)
            if best_s != ref_s:
                assert margin < 1e-5, (i, variant, best_s, ref_s, margin)      # only a near-tie may differ
                amb += 1
                continue
            exact += 1
            worst = max(worst, float(np.max(np.abs(llr - g[key][i]))))
    assert worst <= 1e-5, worst
    rows = np.array(rows)
    report = {"frames_x_variants": int(rows.size), "shift_equal_to_reference": exact, "tie_ambiguous_shift_differs": amb,
              "relative_margin_best_vs_runner_up": {"min": float(rows.min()), "median": float(np.median(rows)),
                                                    "below_1e-5": int((rows < 1e-5).sum()), "below_1e-4": int((rows < 1e-4).sum())},
              "worst_abs_llr_error_vs_reference": worst,
              "note": "256 config-3 windows x PN variants 0/1, oracle vs reference _llr at the reference's first peak"}
    # the committed report is an ASSERTED record of this run, not an output of it (ES_WRITE_H1_REPORT=1 regenerates it)
    path = os.path.join(HERE, "golden", "h1_margin_report.json")
    if os.environ.get("ES_WRITE_H1_REPORT") == "1":
        with open(path, "w") as fh:
            json.dump(report, fh, indent=1)
    with open(path) as fh:
        committed = json.load(fh)
    assert committed["frames_x_variants"] == report["frames_x_variants"]
    assert committed["shift_equal_to_reference"] == exact and committed["tie_ambiguous_shift_differs"] == amb
    assert committed["relative_margin_best_vs_runner_up"]["below_1e-5"] == report["relative_margin_best_vs_runner_up"]["below_1e-5"]
    assert abs(committed["relative_margin_best_vs_runner_up"]["min"] - report["relative_margin_best_vs_runner_up"]["min"]) <= 1e-9
    assert amb == 0 and exact == rows.size          # observed: every chosen shift equals the reference's; the margin rule excuses nothing here


@pytest.mark.parametrize("fs", [44_100, 96_000])
def test_other_fs_target_matches_reference(oracle, fs):
    """WatermarkDetector(fs_target = 44 100 / 96 000): band-pass design, preamble template and matched filter follow the rate (550 taps in the
    18-22 kHz band at 44 100 Hz).  24 C3 windows through the reference built for that rate (oracle/refshim/gen_golden_r3.py fs): tables, sync,
    header decode at the visited peaks, _llr (both PN variants) at the first peak."""
    from echoseal_amd.crypto import SecureChannel
    from echoseal_amd.tables import pack_tables
    g = _g(f"fs{fs}_windows.npz")
    tables = pack_tables(fs)
    ba, tpl, taps, ntaps, _ = tables
    assert int(ntaps.max()) > 160 and int(ntaps.max()) <= 576
    sec = SecureChannel(KEY)
    hdr_pn = sec.pn_bits(0, 128)
    worst = 0.0
    for i in range(g["win"].shape[0]):
        band = int(g["band"][i]); ctr = int(g["ctr"][i])
        assert int(ntaps[band]) == int(g["ntaps"][i])
        y, corr, thr, med, mad, peaks, total, fb = _sync(oracle, tables, g["win"][i], band)
        if f"corr/{i:03d}" in g.files:
            assert np.max(np.abs(corr - g[f"corr/{i:03d}"])) < 1e-12
        assert abs(thr - float(g["thr"][i])) < 1e-12 and fb == bool(g["fallback"][i]) and total == int(g["npeaks"][i])
        k = min(total, 32)
        assert list(peaks[:k]) == list(g["peaks"][i, :k]), i
        h = taps[band, :ntaps[band]]
        visited = [int(p) for p in g["peaks"][i, :min(int(g["npeaks"][i]), 25)] if p + 1215 <= y.size][:5]
        assert len(visited) == min(5, int(g["nvisited"][i]))
        for j, st in enumerate(visited):
            ok, val, score, _ = oracle.decode_header(y[st:st + 1215], hdr_pn, h)
            assert ok == bool(g["hdr"][i, j, 0]) and val == int(g["hdr"][i, j, 1]), (i, j)
            assert abs(score - g["hdr"][i, j, 2]) <= 1e-4 * max(1.0, abs(g["hdr"][i, j, 2]))
        st = int(g["peaks"][i, 0])
        pn = sec.pn_bits(ctr, 1215)
        for variant, key, pnb in ((0, "llr0", pn[191:1215]), (1, "llr1", pn[:1024])):
            llr, best_s, s0, s1 = oracle.llr(y[st:st + 1215], pnb, h)
            if y.size - st <= 191:
                assert not llr.any() and not g[key][i].any()
                continue
            if best_s != int(g["best_s"][i, variant]):
                assert (s0 - s1) / max(abs(s0), 1e-30) < 1e-5, (i, variant)     # only a near-tie may differ
                continue
            worst = max(worst, float(np.max(np.abs(llr - g[key][i]))))
    assert worst <= 1e-5, worst


def test_multi_peak_records_match_reference(oracle, tables):
    from echoseal_amd.crypto import SecureChannel
    g = _g("sync_multi.npz")
    ba, tpl, taps, ntaps, _ = tables
    hdr_pn = SecureChannel(KEY).pn_bits(0, 128)
    several = 0
    for i in range(int(g["count"])):
        t = f"{i:02d}"
        band = int(g[f"{t}/band"])
        y, corr, thr, med, mad, peaks, total, fb = _sync(oracle, tables, g[f"{t}/x"], band)
        assert abs(thr - float(g[f"{t}/thr"])) < 1e-12 and fb == bool(g[f"{t}/fallback"])
        ref = list(g[f"{t}/peaks"])
        assert total == len(ref) and list(peaks[:total]) == ref[:32], t
        several += (len(ref) >= 2 and not fb)
        for j, st in enumerate(g[f"{t}/visited"]):
            ok, val, score, _ = oracle.decode_header(y[st:st + 1215], hdr_pn, taps[band, :ntaps[band]])
            assert ok == bool(g[f"{t}/hdr"][j, 0]) and val == int(g[f"{t}/hdr"][j, 1])
    assert several >= 20


@pytest.mark.parametrize("mode", ["glibc", "default"])
def test_polar_bulk_matches_reference(oracle, mode):
    g = _g(f"polar_bulk_{mode}.npz")
    llrs = _g("polar_bulk_glibc.npz")["llr"]
    kinds = _g("polar_bulk_glibc.npz")["kind"]       # 0 detector-produced, 1 AWGN, 2 tie-heavy, 3 garbage, 4 weak flips
    L = int(g["list_size"])
    n = llrs.shape[0]
    assert n == 1024 and L == 8
    listed = metric_flips = 0
    for i in range(n):
        info, ok, took = oracle.polar_decode(llrs[i], L)
        assert took == bool(g["took_list"][i]), i
        if not took:
            assert ok == bool(g["ok"][i]) and np.array_equal(np.packbits(info), g["info"][i]), i
            continue
        listed += 1
        nn, ci, cm, cc = oracle.scl_list(llrs[i], L)
        same_bits = np.array_equal(np.packbits(ci, axis=1), g["cand_info"][i])
        if mode == "glibc":                     # the C library's exp/log1p: bit for bit
            assert np.array_equal(cm.view(np.uint64), g["cand_metric"][i].view(np.uint64)), i
            assert same_bits and np.array_equal(cc, g["cand_crc"][i]), i
            assert ok == bool(g["ok"][i]) and np.array_equal(np.packbits(info), g["info"][i]), i
        else:                                   # AVX-512 exp/log1p of the build host's NumPy: metrics a few ulp apart,
            if same_bits:                       # which can reorder two candidates whose metrics (nearly) tie
                assert np.allclose(cm, g["cand_metric"][i], rtol=1e-12, atol=0), i
                assert ok == bool(g["ok"][i]) and np.array_equal(np.packbits(info), g["info"][i]), i
            else:                               # an exact tie broken the other way early on: the two runs of the
                metric_flips += 1               # REFERENCE (AVX-512 vs C-library NumPy) end with different lists
                assert kinds[i] == 2, (i, int(kinds[i]))                       # only the tie-heavy family
                assert not np.array_equal(g["cand_info"][i], _g("polar_bulk_glibc.npz")["cand_info"][i])
    assert listed >= 900
    assert metric_flips <= 2, metric_flips


@pytest.mark.parametrize("mode", ["glibc", "default"])
@pytest.mark.parametrize("L", [1, 4, 16])
def test_polar_sweep_matches_reference(oracle, mode, L):
    """BASELINE config 5 sweeps the list size over 1 / 4 / 8 / 16: 256 vectors per size (detector-produced on C3 windows and through
    the config-5 surrogate channel, AWGN, tie-heavy) through the reference's PolarCode.decode (oracle/refshim/gen_golden_r3.py)."""
    g = _g(f"polar_sweep_{mode}.npz")
    gl = _g("polar_sweep_glibc.npz")
    llrs, kinds = gl[f"L{L}/llr"], gl[f"L{L}/kind"]          # 0 C3 detector, 1 lossy-channel detector, 2 AWGN, 3 tie-heavy
    n = llrs.shape[0]
    assert n == 256
    listed = flips = 0
    for i in range(n):
        info, ok, took = oracle.polar_decode(llrs[i], L)
        assert took == bool(g[f"L{L}/took_list"][i]), i
        if not took:
            assert ok == bool(g[f"L{L}/ok"][i]) and np.array_equal(np.packbits(info), g[f"L{L}/info"][i]), i
            continue
        listed += 1
        nn, ci, cm, cc = oracle.scl_list(llrs[i], L)
        assert nn == int(g[f"L{L}/ncand"][i]) == L, i
        same_bits = np.array_equal(np.packbits(ci, axis=1), g[f"L{L}/cand_info"][i])
        if mode == "glibc":
            assert np.array_equal(cm.view(np.uint64), g[f"L{L}/cand_metric"][i].view(np.uint64)), i
            assert same_bits and np.array_equal(cc, g[f"L{L}/cand_crc"][i]), i
            assert ok == bool(g[f"L{L}/ok"][i]) and np.array_equal(np.packbits(info), g[f"L{L}/info"][i]), i
        elif same_bits:
            assert np.allclose(cm, g[f"L{L}/cand_metric"][i], rtol=1e-12, atol=0), i
            assert ok == bool(g[f"L{L}/ok"][i]) and np.array_equal(np.packbits(info), g[f"L{L}/info"][i]), i
        else:                                   # the reference's two NumPy modes break an exact tie differently: tie-heavy rows only
            flips += 1
            assert kinds[i] == 3, (i, int(kinds[i]))
            assert not np.array_equal(g[f"L{L}/cand_info"][i], gl[f"L{L}/cand_info"][i])
    assert listed >= 200
    assert flips <= 2, flips


@pytest.mark.parametrize("name", ["polar_codes", "polar_codes2"])
@pytest.mark.parametrize("mode", ["default", "glibc"])
def test_other_codes_match_reference(oracle, mode, name):
    """PolarCode(1024, K) for K other than the detector's 448 (the reference's class takes any K, rtwm/fastpolar.py:209-234): K = 16, 64, 200,
    512, 1000 at list sizes 1 and 8, six vectors each (clean, noisy, noise only, bit flips at +-12, constant magnitude, all zero) through the
    reference's decode (oracle/refshim/gen_golden_r3.py codes); polar_codes2: K = 9, 13, 301, 1023, 1024 (one information bit; K - 8 not a whole
    number of bytes -- rows are np.packbits of the information bits; no frozen position)."""
    g = _g(f"{name}_{mode}.npz"); gl = _g(f"{name}_glibc.npz")
    listed = 0
    for K in g["ks"]:
        llrs = gl[f"K{K}/llr"].astype(np.float64)
        with oracle.code_k(int(K)):
            for L in g["lists"]:
                t = f"K{K}/L{L}"
                for i in range(llrs.shape[0]):
                    info, ok, took = oracle.polar_decode(llrs[i], int(L))
                    assert took == bool(g[f"{t}/took_list"][i]), (t, i)
                    if not took:
                        assert ok == bool(g[f"{t}/ok"][i]) and np.array_equal(np.packbits(info), g[f"{t}/info"][i]), (t, i)
                        continue
                    listed += 1
                    nn, ci, cm, cc = oracle.scl_list(llrs[i], int(L))
                    assert nn == int(g[f"{t}/ncand"][i]), (t, i)
                    same_bits = np.array_equal(np.packbits(ci, axis=1), g[f"{t}/cand_info"][i][:nn])
                    if mode == "glibc":
                        assert same_bits and np.array_equal(cm.view(np.uint64), g[f"{t}/cand_metric"][i][:nn].view(np.uint64)), (t, i)
                        assert np.array_equal(cc, g[f"{t}/cand_crc"][i][:nn]), (t, i)
                    if same_bits:                  # (NumPy's own exp/log1p may break an exact tie the other way: the all-zero / constant rows)
                        assert np.allclose(cm, g[f"{t}/cand_metric"][i][:nn], rtol=1e-12, atol=0), (t, i)
                        assert ok == bool(g[f"{t}/ok"][i]) and np.array_equal(np.packbits(info), g[f"{t}/info"][i]), (t, i)
                    else:
                        assert i >= 3, (t, i)
    assert oracle.polar_tables()[1].size == 448          # the context manager put the detector's code back
    assert listed >= 30


def _scl_result_from_oracle(oracle, llr, L):
    from echoseal_amd.engine import SclResult
    hinfo, hok = oracle.polar_hard(llr.astype(np.float64))
    nn, ci, cm, cc = oracle.scl_list(llr.astype(np.float64), L)
    t = torch.from_numpy
    return SclResult(t(np.packbits(hinfo)[None]), t(np.array([hok], np.uint8)), t(np.packbits(ci, axis=1)[None]),
                     t(cm[None]), t(cc[None]), t(np.array([nn], np.int32)))


def _validator(spec, arg, ctr_key, seen):
    from echoseal_amd.crypto import SecureChannel
    sec = SecureChannel(KEY)

    def v(payload):
        seen.append(bytes(payload))
        if spec == "reject":
            return False
        if spec == "call":
            return len(seen) == int(arg)
        if spec == "raise":
            raise RuntimeError("validator failure")
        if spec == "payload":
            return bytes(payload) == bytes(arg.tobytes())
        try:
            pt = sec.open(payload)
        except Exception:
            return False
        return pt.startswith(b"ESAL") and int.from_bytes(pt[4:8], "big") == int(arg)
    return v


def test_decode_with_validator_matches_reference(oracle):
    """Host tail of PolarCode.decode (engine.select_payload: rtwm/fastpolar.py:268-276, 332-359) over the oracle's
    candidates: result AND the exact sequence of payloads shown to the validator equal the reference's."""
    from echoseal_amd.engine import select_payload
    g = _g("polar_validator.npz")
    n = int(g["count"])
    accepted = listed = 0
    for i in range(n):
        t = f"{i:03d}"
        L = int(g[f"{t}/L"]); spec = str(g[f"{t}/spec"]); arg = g[f"{t}/arg"]
        res = _scl_result_from_oracle(oracle, g[f"{t}/llr"], L)
        seen = []
        payload, ok = select_payload(res, 0, _validator(spec, arg, KEY, seen))
        assert ok == bool(g[f"{t}/ok"]), (i, spec)
        assert payload == g[f"{t}/info"].tobytes(), (i, spec)
        ref_seen = [r.tobytes() for r in g[f"{t}/seen"]]
        assert seen == ref_seen, (i, spec, len(seen), len(ref_seen))
        accepted += ok; listed += len(ref_seen) >= 2
    assert n >= 280 and accepted >= 25 and listed >= 30


def test_oracle_select_validated_matches_reference(oracle):
    """The oracle's C selection with the detector's AEAD validator (what es_select_batch is checked against on the GPU)."""
    from echoseal_amd.crypto import SecureChannel
    g = _g("polar_validator.npz")
    key = SecureChannel(KEY)._aead._key
    done = 0
    for i in range(int(g["count"])):
        t = f"{i:03d}"
        if str(g[f"{t}/spec"]) != "aead":
            continue
        L = int(g[f"{t}/L"]); llr = g[f"{t}/llr"].astype(np.float64)
        hinfo, hok = oracle.polar_hard(llr)
        nn, ci, cm, cc = oracle.scl_list(llr, L)
        payload, ok, which = oracle.select_validated(key, int(g[f"{t}/arg"]), np.packbits(hinfo), hok, np.packbits(ci, axis=1), cc, cm, nn)
        assert ok == int(bool(g[f"{t}/ok"])) and payload == g[f"{t}/info"].tobytes(), i
        done += 1
    assert done >= 90


@pytest.mark.parametrize("mode", ["glibc", "default"])
def test_list_sizes_that_are_not_powers_of_two(oracle, mode):
    """The reference takes any list_size >= 1 (rtwm/fastpolar.py:215); fixtures for 3, 5, 6, 12, 24, 100."""
    g = _g(f"polar_odd_{mode}.npz")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/llr")})
    checked = 0
    for name in names:
        llr = g[f"{name}/llr"]
        for L in (3, 5, 6, 12, 24, 100):
            info, ok, took = oracle.polar_decode(llr, L)
            assert ok == bool(g[f"{name}/L{L}/ok"]) and np.array_equal(np.packbits(info), g[f"{name}/L{L}/info"]), (name, L)
            key = f"{name}/L{L}/cand_metric"
            if key in g.files:
                n, ci, cm, cc = oracle.scl_list(llr, L)
                assert n == L and np.array_equal(np.packbits(ci, axis=1), g[f"{name}/L{L}/cand_info"]), (name, L)
                assert np.array_equal(cm, g[key]) if mode == "glibc" else np.allclose(cm, g[key], rtol=1e-12, atol=0)
                assert np.array_equal(cc, g[f"{name}/L{L}/cand_crc"])
                checked += 1
    assert checked >= 20
