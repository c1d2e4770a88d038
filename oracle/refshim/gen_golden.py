#!/usr/bin/env python3
"""Capture golden vectors from the reference itself (build container only).

    python -m oracle.refshim.gen_golden            # writes tests/golden/*.npz

The reference (/root/reference, read-only) is imported through oracle/refshim/shim.py and RUN;
only its inputs and outputs are stored -- no reference source.  Two NumPy run-time configurations
are captured for the polar decoder, because NumPy's float64 exp/log1p (used by the path-metric
penalty, rtwm/fastpolar.py:36) dispatch to AVX-512/SVML kernels on this CPU and to the C library
otherwise:
    polar_default.npz   NumPy as installed (AVX-512 dispatch active on the build host)
    polar_glibc.npz     NPY_DISABLE_CPU_FEATURES=AVX512* -> libm exp/log1p (what any non-AVX-512
                        host computes); our oracle matches THESE metrics bit for bit.
Detector vectors (sync, LLR, header) do not depend on that switch.
"""
from __future__ import annotations

import contextlib
import io
import os
import subprocess
import sys
import types

AVX512_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
LISTS = (1, 4, 8, 16, 32)


def polar_cases(np, ref_polar_fast, det_llrs):
    """name -> LLR vector (float32 or float64 exactly as a caller would pass it)."""
    enc = ref_polar_fast.encode
    cases = {}
    rng = np.random.default_rng(7)
    cases["garbage_rng7"] = np.clip(rng.normal(0.0, 2.0, 1024), -12, 12)
    cases["zeros"] = np.zeros(1024)
    cw = enc(bytes(range(55)))
    cases["clean10_range55"] = np.where(cw == 1, 10.0, -10.0).astype(np.float32)
    flip = cw.copy(); flip[[5, 300, 900]] ^= 1
    cases["clean10_3err"] = np.where(flip == 1, 10.0, -10.0).astype(np.float32)
    cases["clean2_A55"] = (2.0 * (2.0 * enc(b"A" * 55).astype(np.float32) - 1.0)).astype(np.float32)
    for seed in (1234, 4321):                               # tests/test_polar.py:64-109
        r = np.random.default_rng(seed)
        if seed == 1234:
            info = r.integers(0, 2, 440, dtype=np.uint8)
            code = enc(np.packbits(info).tobytes())
        else:
            code = enc(r.integers(0, 256, size=55, dtype=np.uint8).tobytes())
        rx = 2.0 * code.astype(np.float64) - 1.0 + r.normal(0.0, 0.15, 1024)
        cases[f"awgn015_seed{seed}"] = 2.0 * rx / 0.15 ** 2
    r = np.random.default_rng(99)
    code = enc(r.integers(0, 256, size=55, dtype=np.uint8).tobytes())
    for sigma in (0.5, 0.7, 0.9, 1.1):
        rx = 2.0 * code.astype(np.float64) - 1.0 + r.normal(0.0, sigma, 1024)
        cases[f"awgn{int(sigma * 100):03d}"] = np.clip(2.0 * rx / sigma ** 2, -12, 12).astype(np.float32)
    cases["neg_garbage"] = -cases["garbage_rng7"]
    cases["ties_pm12"] = np.where(np.random.default_rng(5).integers(0, 2, 1024) == 1, 12.0, -12.0)
    for k, v in det_llrs.items():
        cases[k] = v
    return cases


def run_polar(np, fp, cases):
    """Run reference PolarCode.decode for every case / list size, capturing the final list."""
    import builtins
    captured = {}

    def spy_sorted(seq, key=None):
        out = builtins.sorted(seq, key=key)
        if seq and hasattr(seq[0], "metric"):
            captured["metric"] = np.array([p.metric for p in out], dtype=np.float64)
            captured["u"] = np.stack([p.u.copy() for p in out])
        return out

    fp.sorted = spy_sorted                     # module-global lookup precedes builtins
    out = {}
    for name, llr in cases.items():
        out[f"{name}/llr"] = llr
        for L in LISTS:
            pc = fp.PolarCode(1024, 448, list_size=L, crc_size=8)
            captured.clear()
            bits, ok = pc.decode(llr)
            out[f"{name}/L{L}/info"] = np.packbits(bits)
            out[f"{name}/L{L}/ok"] = np.array(ok)
            if captured:
                data = captured["u"][:, pc._data_pos]
                out[f"{name}/L{L}/cand_info"] = np.packbits(data[:, :440], axis=1)
                out[f"{name}/L{L}/cand_metric"] = captured["metric"]
                out[f"{name}/L{L}/cand_crc"] = np.array(
                    [pc._crc_ok(d[:440], d[440:448]) for d in data], dtype=np.uint8)
            print(f"  polar {name} L={L} ok={ok} list={'yes' if captured else 'no'}", flush=True)
    del fp.sorted
    return out


def run_detector(np, rtwm):
    """Frames from the reference embedder (frozen payload) through the reference detector stages."""
    from scipy.signal import lfilter, correlate
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.detector import WatermarkDetector, FRAME_LEN
    from rtwm.utils import choose_band, butter_bandpass, BAND_PLAN
    sink = io.StringIO()
    out = {}
    det_llrs = {}
    idx = 0
    for key_name, key in (("AA", b"\xAA" * 32), ("00", b"\x00" * 32)):
        with contextlib.redirect_stdout(sink):
            tx = WatermarkEmbedder(key)
            rx = WatermarkDetector(key, list_size=8)
        out[f"static/{key_name}/hdr_pn"] = np.packbits(rx.sec.pn_bits(0, 128))
        for band_i, band in enumerate(BAND_PLAN):
            with contextlib.redirect_stdout(sink):
                out[f"static/taps{band_i}"] = rx._matched_filter_taps(band)
            b, a = butter_bandpass(*band, 48000, order=4)
            out[f"static/ba{band_i}"] = np.concatenate((b, a))
        prng = np.random.default_rng(2024)
        for ctr in (0, 1, 2, 3, 5, 255, 1024):
            payload = bytes(range(55)) if ctr == 5 else prng.integers(0, 256, 55, dtype=np.uint8).tobytes()
            with contextlib.redirect_stdout(sink):
                tx.frame_ctr = ctr
                tx._build_payload = types.MethodType(lambda s, p=payload: p, tx)
                frame = tx._make_frame_chips()
            for noise_name, sigma in (("clean", 0.0), ("noisy", 0.25)):
                if sigma and ctr not in (1, 3, 255):
                    continue
                x = frame if not sigma else (frame + np.random.default_rng(1000 + ctr).normal(0, sigma, frame.size)).astype(np.float32)
                band = choose_band(key, ctr)
                b, a = butter_bandpass(*band, 48000, order=4)
                with contextlib.redirect_stdout(sink):
                    y = lfilter(b, a, x.astype(np.float32, copy=False))                 # detector.py:59-60
                    tpl = lfilter(b, a, lfilter(b, a, rx._pre_sy))                       # :67-69
                    tpl = tpl / float(np.sqrt(np.sum(tpl * tpl)) + 1e-12)
                    e_y = np.sqrt(np.convolve(y * y, np.ones(63, dtype=np.float32), mode="valid")) + 1e-12
                    corr = correlate(y, tpl, mode="valid") / e_y                         # :76-79
                    med = float(np.median(corr)); mad = float(np.median(np.abs(corr - med))) + 1e-12
                    thr = min(med + 4.5 * 1.4826 * mad, 0.95)                            # :83-86
                    peaks = []
                    for i in range(corr.size):                                           # :89-96
                        if corr[i] < thr:
                            continue
                        lo = max(0, i - 607); hi = min(corr.size, i + 608)
                        if corr[i] >= corr[lo:hi].max():
                            peaks.append(i)
                    fallback = not peaks
                    if not peaks:
                        peaks = [int(v) for v in np.argsort(corr)[-min(5, corr.size):][::-1]]
                    llr0 = rx._llr(y, ctr, 0); llr1 = rx._llr(y, ctr, 1)               # :296-416
                    llr_short = rx._llr(y[:700], ctr, 0)
                    hdr = rx._decode_header(y, band)                                     # :452-515
                log = sink.getvalue()
                best = [int(l.split("best_s=")[1].split(",")[0]) for l in log.splitlines() if "best_s=" in l][-3:]
                sink.seek(0); sink.truncate()
                tag = f"det/{idx:02d}"
                out[f"{tag}/key"] = np.frombuffer(key, np.uint8)
                out[f"{tag}/ctr"] = np.array(ctr)
                out[f"{tag}/band"] = np.array(BAND_PLAN.index(band))
                out[f"{tag}/payload"] = np.frombuffer(payload, np.uint8)
                out[f"{tag}/x"] = x
                out[f"{tag}/y"] = y
                out[f"{tag}/tpl"] = tpl
                out[f"{tag}/corr"] = corr
                out[f"{tag}/thr"] = np.array(thr)
                out[f"{tag}/peaks"] = np.array(peaks, dtype=np.int32)
                out[f"{tag}/fallback"] = np.array(fallback)
                out[f"{tag}/llr0"] = llr0
                out[f"{tag}/llr1"] = llr1
                out[f"{tag}/llr_short700"] = llr_short
                out[f"{tag}/best_s"] = np.array(best, dtype=np.int32)     # variant0, variant1, short
                out[f"{tag}/hdr"] = np.array([float(hdr[0]), float(hdr[1]), float(hdr[2])])
                out[f"{tag}/pn"] = np.packbits(rx.sec.pn_bits(ctr, FRAME_LEN))
                if key_name == "AA" and ctr in (0, 3, 255):
                    det_llrs[f"det_{noise_name}_ctr{ctr}"] = llr0
                print(f"  detector key={key_name} ctr={ctr} {noise_name} peaks={peaks[:3]} best_s={best} hdr={hdr}", flush=True)
                idx += 1
    out["det/count"] = np.array(idx)
    return out, det_llrs


def main() -> None:
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode == "all":
        os.makedirs(GOLD, exist_ok=True)
        env = dict(os.environ, PYTHONPATH=ROOT)
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden", "default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden", "glibc"], cwd=ROOT, env=env)
        return
    import numpy as np
    from oracle.refshim.shim import load_reference
    rtwm = load_reference()
    import rtwm.fastpolar as fp
    import rtwm.polar_fast as pf
    det, det_llrs = run_detector(np, rtwm)
    if mode == "default":
        np.savez_compressed(os.path.join(GOLD, "detector.npz"), **det)
    pol = run_polar(np, fp, polar_cases(np, pf, det_llrs))
    pol["meta/numpy"] = np.array(np.__version__)
    pol["meta/mode"] = np.array(mode)
    np.savez_compressed(os.path.join(GOLD, f"polar_{mode}.npz"), **pol)


if __name__ == "__main__":
    main()
