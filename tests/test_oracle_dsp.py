"""Pin the C oracle's sync / LLR stages (oracle/c/eso_dsp.c) against the reference's outputs
(tests/golden/detector.npz, captured by oracle/refshim/gen_golden.py) and against SciPy/NumPy."""
import numpy as np
import pytest
from scipy.signal import lfilter

from echoseal_amd.tables import band_coeffs, matched_filter_taps, pack_tables, preamble_template
from echoseal_amd.utils import BAND_PLAN


def _frames(g):
    return [f"det/{i:02d}" for i in range(int(g["det/count"]))]


def test_tables_match_reference(golden_detector):
    g = golden_detector
    ba, tpl, taps, ntaps, _ = pack_tables()
    assert list(ntaps) == [131, 116, 123, 93]
    for i in range(4):
        assert np.array_equal(ba[i], g[f"static/ba{i}"])
        assert np.array_equal(taps[i, :ntaps[i]], g[f"static/taps{i}"])
    for t in _frames(g):
        assert np.array_equal(tpl[int(g[f"{t}/band"])], g[f"{t}/tpl"])


def test_lfilter_is_bit_exact_with_scipy(oracle):
    rng = np.random.default_rng(1)
    for band in BAND_PLAN:
        b, a = band_coeffs(band, 48_000)
        for n in (1, 8, 63, 1215, 5000):
            x = rng.normal(0, 0.3, n).astype(np.float32)
            assert np.array_equal(oracle.lfilter(b, a, x), lfilter(b, a, x))


def test_pairwise_sum_is_numpy(oracle):
    rng = np.random.default_rng(2)
    for n in (1, 5, 7, 8, 9, 100, 128, 129, 255, 256, 257, 959, 966, 978, 1000, 1023, 1024):
        a = rng.normal(0, 1, n).astype(np.float32)
        assert oracle.sum_f32(a) == np.add.reduce(a), n


def test_sync_matches_reference(oracle, golden_detector):
    g = golden_detector
    for t in _frames(g):
        band = int(g[f"{t}/band"])
        b, a = band_coeffs(BAND_PLAN[band], 48_000)
        y = oracle.lfilter(b, a, g[f"{t}/x"])
        assert np.array_equal(y, g[f"{t}/y"]), t                              # bit exact
        corr = oracle.ncc(y, g[f"{t}/tpl"])
        assert np.max(np.abs(corr - g[f"{t}/corr"])) < 1e-12, t               # BLAS-order noise only
        thr, _, _ = oracle.cfar_threshold(corr)
        assert abs(thr - float(g[f"{t}/thr"])) < 1e-12
        peaks, total, fb = oracle.pick_peaks(corr, thr)
        assert fb == bool(g[f"{t}/fallback"])
        assert list(peaks[:total]) == list(g[f"{t}/peaks"]), t               # sync offsets: exact


def test_llr_matches_reference(oracle, golden_detector):
    g = golden_detector
    worst = 0.0
    for t in _frames(g):
        band = int(g[f"{t}/band"])
        h = matched_filter_taps(BAND_PLAN[band], 48_000)
        pn = np.unpackbits(g[f"{t}/pn"])[:1215]
        y = g[f"{t}/y"]
        for variant, key, pnb in ((0, "llr0", pn[191:1215]), (1, "llr1", pn[:1024])):
            llr, best_s, s0, s1 = oracle.llr(y, pnb, h)
            assert best_s == int(g[f"{t}/best_s"][variant]), (t, variant)
            worst = max(worst, float(np.max(np.abs(llr - g[f"{t}/{key}"]))))
        llr, best_s, _, _ = oracle.llr(y[:700], pn[191:1215], h)              # short frame, zero padded
        assert best_s == int(g[f"{t}/best_s"][2])
        worst = max(worst, float(np.max(np.abs(llr - g[f"{t}/llr_short700"]))))
        assert not llr[700 - 191:].any()
    assert worst <= 1e-5, worst                                               # north-star tolerance


def test_llr_degenerate_frames(oracle):
    h = matched_filter_taps(BAND_PLAN[0], 48_000)
    pn = np.zeros(1024, np.uint8)
    for n in (0, 100, 191):                                                   # detector.py:320-325
        llr, *_ = oracle.llr(np.ones(n), pn, h)
        assert not llr.any()
    llr, best_s, s0, _ = oracle.llr(np.zeros(1215), pn, h)                    # silent frame
    assert not llr.any() and best_s == -130 and s0 == 0.0


def test_template_is_unit_norm():
    for band in BAND_PLAN:
        tpl = preamble_template(band, 48_000)
        assert tpl.shape == (63,) and abs(float(np.sum(tpl * tpl)) - 1.0) < 1e-9
