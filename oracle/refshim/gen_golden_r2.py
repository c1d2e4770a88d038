#!/usr/bin/env python3
"""Round-2 golden vectors, captured by RUNNING the reference (build container only; /root/reference is read-only
and never travels).  Only inputs and outputs are stored under tests/golden/ -- no reference source.

    python -m oracle.refshim.gen_golden_r2 [all|c3|multi|verify|bulk|validator]

    c3_windows.npz      256 BASELINE-config-3 windows (W = 2048, +-5 % resampling, -15 dB AWGN; generator
                        echoseal_amd.workloads.c3_windows) through the reference's own
                        WatermarkDetector._scan_band_multi_frame (rtwm/detector.py:56-152): thr / median / MAD, the
                        correlation row, the peak list, the fallback flag, the header decode at every peak the
                        function visits, and _llr (variants 0 and 1) at the first peak.
    sync_multi.npz      records with several peaks: 3-frame clips in one band (+ noise) and frame-sized records whose
                        correlation row has a second peak, same capture.
    verify3s.npz        a 3 s noisy clip through WatermarkDetector.verify() (list size 1): every (band, peak, ctr) try in
                        order, the header decode of every peak, the result.
    polar_bulk.npz      1 024 LLR vectors (AWGN sigma 0.3-1.1, detector-produced = the c3 LLRs, tie-heavy, garbage, wide
                        range float64) through PolarCode.decode(list_size=8): (info, ok), final list (bits, metrics, CRC
                        flags) in both NumPy run-time modes (see gen_golden.py: `default` / `glibc`).
    polar_odd_*.npz     PolarCode.decode with list sizes that are not powers of two (3, 5, 6, 12, 24, 100), both NumPy modes.
    polar_validator.npz PolarCode.decode WITH a validator (rtwm/fastpolar.py:268-276, 335-359): reject-all, accept the
                        k-th call, raising validator, accept-one-payload, and the detector's own AEAD closure
                        (rtwm/detector.py:168-176) with the right and a wrong counter; records every payload the
                        validator was shown, in order.

How the sync values are captured from the REAL function rather than a restatement: the module global `np` of
rtwm.detector is replaced by a proxy that forwards everything to NumPy and records the arguments / results of
np.max (receives the correlation row), np.median (median, then MAD) and np.argsort (fallback); _decode_header and
_try_decode_frame are wrapped to record the frame start (offset of the view inside y) and, for the latter, to return
False without decoding.  The full peak list (the function only visits peaks whose frame fits, at most 25) is recomputed
from the captured row by the rule of rtwm/detector.py:89-99 and ASSERTED against what the function reveals (peak count,
its first five peak values from stdout, the starts it visits).
"""
from __future__ import annotations

import contextlib
import io
import os
import re
import subprocess
import sys
import types

AVX512_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
KEY = b"\xAA" * 32


# ----------------------------------------------------------------------------------------------- capture helpers
class NpSpy:
    """Stands in for the `np` global of rtwm.detector: NumPy itself, with three calls recorded."""

    def __init__(self, np):
        self._np = np
        self.reset()

    def reset(self):
        self.rows, self.medians, self.argsorts = [], [], 0

    def __getattr__(self, name):
        return getattr(self._np, name)

    def max(self, a, *args, **kw):
        if not args and not kw and getattr(a, "ndim", 0) == 1 and a.dtype == self._np.float64:
            self.rows.append(a)
        return self._np.max(a, *args, **kw)

    def median(self, a, *args, **kw):
        r = self._np.median(a, *args, **kw)
        self.medians.append(float(r))
        return r

    def argsort(self, a, *args, **kw):
        self.argsorts += 1
        return self._np.argsort(a, *args, **kw)


def scan_capture(np, det_mod, rx, signal, band):
    """Run the reference's _scan_band_multi_frame(signal, band) and return what it computed."""
    spy = NpSpy(np)
    visited, headers = [], []
    orig_hdr = type(rx)._decode_header

    def hdr_spy(self, frame, b):
        visited.append((frame.__array_interface__["data"][0] - ys[0].__array_interface__["data"][0]) // 8)
        spy_medians = len(spy.medians)
        r = orig_hdr(self, frame, b)
        del spy.medians[spy_medians:]
        headers.append((float(r[0]), float(r[1]), float(r[2])))
        return r

    ys = []
    saved_lfilter = det_mod.lfilter

    def lfilter_spy(*a, **kw):
        r = saved_lfilter(*a, **kw)
        ys.append(r)                                   # first call = y (rtwm/detector.py:60)
        return r

    sink = io.StringIO()
    saved_np = det_mod.np
    det_mod.np = spy
    det_mod.lfilter = lfilter_spy
    rx._decode_header = types.MethodType(hdr_spy, rx)
    rx._try_decode_frame = types.MethodType(lambda self, frame, ctr: False, rx)
    try:
        with contextlib.redirect_stdout(sink):
            res = rx._scan_band_multi_frame(signal, band)
    finally:
        det_mod.np = saved_np
        det_mod.lfilter = saved_lfilter
        del rx._decode_header, rx._try_decode_frame
    assert res is False
    log = sink.getvalue()
    if not spy.rows:                                   # record shorter than the template
        return None
    corr = spy.rows[0]
    med, mad_raw = spy.medians[0], spy.medians[1]
    mad = mad_raw + 1e-12
    thr = min(med + 4.5 * 1.4826 * mad, 0.95)                                       # rtwm/detector.py:83-86
    peaks = []
    for i in range(corr.size):                                                      # :89-96
        if corr[i] < thr:
            continue
        lo = max(0, i - 607); hi = min(corr.size, i + 608)
        if corr[i] >= corr[lo:hi].max():
            peaks.append(i)
    fallback = not peaks
    if fallback:
        peaks = [int(v) for v in np.argsort(corr)[-min(5, corr.size):][::-1]]
    # what the real function reveals must agree with the recomputed list
    assert fallback == (spy.argsorts > 0), "fallback branch differs"
    m = re.search(r"\[SCAN\] Found (\d+) peaks", log)
    assert m and int(m.group(1)) == len(peaks), (m and m.group(1), len(peaks))
    m = re.search(r"First 5 peak values: \[(.*)\]", log)
    vals = [float(v) for v in re.findall(r"np\.float64\(([^)]*)\)", m.group(1))] if m else []
    assert vals == [float(corr[p]) for p in peaks[:5]], "first peak values differ"
    want_visits = [p for p in peaks[:25] if p + 1215 <= signal.size]
    assert visited == want_visits, (visited, want_visits)
    m = re.search(r"Threshold: ([0-9.eE+-]+), median: ([0-9.eE+-]+)", log)
    assert m and abs(float(m.group(1)) - thr) < 6e-4
    return {"corr": corr.copy(), "thr": thr, "med": med, "mad": mad, "peaks": np.array(peaks, np.int32),
            "fallback": fallback, "visited": np.array(visited, np.int32),
            "hdr": np.array(headers, np.float64).reshape(-1, 3)}


def best_s_from(log: str):
    return [int(l.split("best_s=")[1].split(",")[0]) for l in log.splitlines() if "best_s=" in l]


def ref_frames(np, ctrs, seed=20260101):
    """Frames from the REFERENCE embedder with the benchmark's sealed payloads (echoseal_amd.embedder.synthetic_payloads
    gives the payload bytes; the frame itself is made by the reference's _make_frame_chips)."""
    from rtwm.embedder import WatermarkEmbedder
    from echoseal_amd.crypto import SecureChannel
    from echoseal_amd.embedder import synthetic_payloads
    sink = io.StringIO()
    payloads = synthetic_payloads(SecureChannel(KEY), ctrs, seed)
    out = []
    with contextlib.redirect_stdout(sink):
        tx = WatermarkEmbedder(KEY)
        for c, p in zip(ctrs, payloads):
            tx.frame_ctr = int(c)
            tx._build_payload = types.MethodType(lambda s, p=p: p, tx)
            out.append(tx._make_frame_chips())
    return np.stack(out).astype(np.float32), payloads


# ----------------------------------------------------------------------------------------------- c3 windows
def gen_c3(n=256):
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.detector as det_mod
    from rtwm.detector import WatermarkDetector
    from rtwm.utils import choose_band, BAND_PLAN
    from echoseal_amd.workloads import c3_windows
    ctrs = list(range(n))
    frames, payloads = ref_frames(np, ctrs)
    win, offs, facs, lens = c3_windows(frames)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        rx = WatermarkDetector(KEY, list_size=8)
    out = {"win": win, "offset": offs, "factor": facs, "length": lens, "ctr": np.array(ctrs, np.int64)}
    bands, thr, med, mad, fb, npk, peaks, nvis, hdrs, llr0, llr1, bs, corr_sha = [], [], [], [], [], [], [], [], [], [], [], [], []
    corr_rows = {}
    for i in range(n):
        band = choose_band(KEY, ctrs[i])
        cap = scan_capture(np, det_mod, rx, win[i], band)
        bands.append(BAND_PLAN.index(band))
        thr.append(cap["thr"]); med.append(cap["med"]); mad.append(cap["mad"]); fb.append(cap["fallback"])
        pk = np.full(32, -1, np.int32); k = min(32, cap["peaks"].size); pk[:k] = cap["peaks"][:k]
        peaks.append(pk); npk.append(cap["peaks"].size)
        h = np.zeros((5, 3)); v = cap["hdr"][:5]; h[:v.shape[0]] = v
        hdrs.append(h); nvis.append(cap["visited"].size)
        if i < 16:
            corr_rows[f"corr/{i:03d}"] = cap["corr"]
        # _llr at the first peak the function visits (or at the first peak, frame possibly short)
        start = int(cap["peaks"][0])
        from scipy.signal import lfilter
        from rtwm.utils import butter_bandpass
        b, a = butter_bandpass(*band, 48000, order=4)
        y = lfilter(b, a, win[i].astype(np.float32, copy=False))
        with contextlib.redirect_stdout(sink):
            sink.seek(0); sink.truncate()
            l0 = rx._llr(y[start:start + 1215], ctrs[i], 0)
            l1 = rx._llr(y[start:start + 1215], ctrs[i], 1)
        b_s = best_s_from(sink.getvalue())
        llr0.append(l0); llr1.append(l1); bs.append((b_s + [0, 0])[:2] if len(b_s) >= 2 else [9999, 9999])
        print(f"  c3 {i}: band {bands[-1]} off {offs[i]} thr {cap['thr']:.4f} peaks {cap['peaks'][:4]} fb {cap['fallback']} "
              f"visited {cap['visited'][:3]} best_s {bs[-1]}", flush=True)
    out.update(band=np.array(bands, np.uint8), thr=np.array(thr), med=np.array(med), mad=np.array(mad),
               fallback=np.array(fb), npeaks=np.array(npk, np.int32), peaks=np.stack(peaks), nvisited=np.array(nvis, np.int32),
               hdr=np.stack(hdrs), llr0=np.stack(llr0), llr1=np.stack(llr1), best_s=np.array(bs, np.int32), **corr_rows)
    np.savez_compressed(os.path.join(GOLD, "c3_windows.npz"), **out)


# ----------------------------------------------------------------------------------------------- multi-peak records
def gen_multi():
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.detector as det_mod
    from rtwm.detector import WatermarkDetector
    from rtwm.utils import choose_band, BAND_PLAN
    from echoseal_amd.utils import band_index
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        rx = WatermarkDetector(KEY, list_size=8)
    rng = np.random.default_rng(77)
    recs = []
    # (a) three frames of ONE band back to back with gaps, clean and noisy: 2-3 peaks per record
    by_band = {b: [c for c in range(400) if band_index(KEY, c) == b] for b in range(4)}
    for b in range(4):
        for rep in range(4):
            cs = by_band[b][rep * 3: rep * 3 + 3]
            fr, _ = ref_frames(np, cs)
            gaps = rng.integers(0, 300, 4)
            x = np.concatenate([np.zeros(gaps[0], np.float32), fr[0], np.zeros(gaps[1], np.float32), fr[1],
                                np.zeros(gaps[2], np.float32), fr[2], np.zeros(gaps[3], np.float32)])
            sigma = (0.0, 0.02, 0.1, 0.3)[rep]
            if sigma:
                x = (x + rng.normal(0, sigma, x.size)).astype(np.float32)
            recs.append((x, b, cs[0]))
    # (b) frame-sized clean records with more than one peak (found with the CPU oracle over ctr 0..4095; kept if the
    #     reference agrees that there are several)
    from oracle import oracle as O
    from echoseal_amd.tables import pack_tables
    ba, tpl, taps, ntaps, _ = pack_tables()
    cand = []
    frames, _ = ref_frames(np, list(range(1024)))
    for c in range(1024):
        b = band_index(KEY, c)
        y = O.lfilter(ba[b][:9], ba[b][9:], frames[c])
        corr = O.ncc(y, tpl[b])
        thr, _, _ = O.cfar_threshold(corr)
        pk, tot, fbk = O.pick_peaks(corr, thr)
        if tot >= 2 and not fbk:
            cand.append(c)
    print(f"  frame-sized records with >= 2 peaks among ctr 0..1023: {len(cand)} -> {cand[:24]}", flush=True)
    for c in cand[:24]:
        recs.append((frames[c], band_index(KEY, c), c))
    out = {"count": np.array(len(recs))}
    for i, (x, b, c) in enumerate(recs):
        cap = scan_capture(np, det_mod, rx, x, BAND_PLAN[b])
        out[f"{i:02d}/x"] = x; out[f"{i:02d}/band"] = np.array(b, np.uint8); out[f"{i:02d}/ctr"] = np.array(c)
        for k in ("thr", "med", "mad", "peaks", "fallback", "visited", "hdr"):
            out[f"{i:02d}/{k}"] = np.asarray(cap[k])
        print(f"  multi {i}: T {x.size} band {b} thr {cap['thr']:.4f} peaks {cap['peaks'][:6]} visited {cap['visited'][:6]}", flush=True)
    np.savez_compressed(os.path.join(GOLD, "sync_multi.npz"), **out)


# ----------------------------------------------------------------------------------------------- verify() on a 3 s clip
def gen_verify():
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.detector import WatermarkDetector
    from oracle.refshim.gen_golden_verify import parse
    sink = io.StringIO()
    rng = np.random.default_rng(31337)
    payloads = {}
    with contextlib.redirect_stdout(sink):
        tx = WatermarkEmbedder(KEY)
        tx._session_nonce = bytes(range(8))

        def fixed_payload(self):
            p = rng.integers(0, 256, 55, dtype=np.uint8).tobytes()
            payloads[self.frame_ctr] = p
            return p
        tx._build_payload = types.MethodType(fixed_payload, tx)
        host = (0.05 * rng.normal(0, 1, 144000)).astype(np.float32)              # 3 s of noise as the host signal
        clip = tx.process(host)
    hdr_log = []
    with contextlib.redirect_stdout(sink):
        rx = WatermarkDetector(KEY, list_size=1)
    orig_hdr = type(rx)._decode_header

    def hdr_spy(self, frame, b):
        r = orig_hdr(self, frame, b)
        hdr_log.append((float(r[0]), float(r[1]), float(r[2])))
        return r
    rx._decode_header = types.MethodType(hdr_spy, rx)
    sink.seek(0); sink.truncate()
    with contextlib.redirect_stdout(sink):
        res = rx.verify(clip, 48000)
    log = sink.getvalue()
    trace = parse(log)
    # peaks visited, in order, with their band: walk the log once more
    vis, band = [], -1
    for line in log.splitlines():
        m = re.match(r"\[SCAN\] Band \((\d+), (\d+)\)", line)
        if m:
            band = int(m.group(1))
        m = re.match(r"\s+Peak@(\d+):.*trying (\d+) counters", line)
        if m:
            vis.append((band, int(m.group(1)), int(m.group(2))))
    out = {"clip": clip.astype(np.float32), "result": np.array(res), "trace": trace,
           "peaks_visited": np.array(vis, np.int64).reshape(-1, 3), "hdr": np.array(hdr_log).reshape(-1, 3),
           "payload_ctrs": np.array(sorted(payloads), np.int64),
           "payloads": np.stack([np.frombuffer(payloads[c], np.uint8) for c in sorted(payloads)])}
    print(f"verify(3 s clip) -> {res}; {trace.shape[0]} tries over {len(vis)} peaks", flush=True)
    np.savez_compressed(os.path.join(GOLD, "verify3s.npz"), **out)


# ----------------------------------------------------------------------------------------------- polar bulk
def bulk_inputs(np):
    """1 024 LLR vectors; float32 unless noted.  Deterministic (seeded), regenerated identically in both modes."""
    from echoseal_amd.polar_fast import encode
    rng = np.random.default_rng(20262)
    rows, kind = [], []
    c3 = np.load(os.path.join(GOLD, "c3_windows.npz"))
    for i in range(256):                                             # detector-produced (reference _llr on C3 windows)
        rows.append(c3["llr0"][i].astype(np.float32)); kind.append(0)
    for i in range(320):                                             # AWGN on random codewords, sigma 0.3 .. 1.1
        sigma = (0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 0.35)[i % 10]
        code = encode(rng.integers(0, 256, 55, dtype=np.uint8).tobytes())
        rx = 2.0 * code.astype(np.float64) - 1.0 + rng.normal(0, sigma, 1024)
        rows.append(np.clip(2.0 * rx / sigma ** 2, -12, 12).astype(np.float32)); kind.append(1)
    for i in range(192):                                             # tie-heavy: few distinct magnitudes, zeros, +-12
        levels = ((0.0, 12.0), (1.0,), (0.5, 1.0, 1.5), (0.0, 2.0, 4.0), (12.0,), (0.0, 0.25))[i % 6]
        mag = rng.choice(np.array(levels), 1024)
        sign = rng.integers(0, 2, 1024) * 2.0 - 1.0
        v = (mag * sign).astype(np.float32)
        if i % 12 >= 6:                                              # partly a real codeword at one magnitude
            code = encode(rng.integers(0, 256, 55, dtype=np.uint8).tobytes())
            flips = rng.random(1024) < 0.04
            v = ((2.0 * (code ^ flips) - 1.0) * levels[-1]).astype(np.float32) if levels[-1] else v
        rows.append(v); kind.append(2)
    for i in range(128):                                             # garbage
        rows.append(np.clip(rng.normal(0, (0.5, 2.0, 6.0, 20.0)[i % 4], 1024), -12, 12).astype(np.float32)); kind.append(3)
    for i in range(128):                                             # codeword with a few weak / flipped positions
        code = encode(rng.integers(0, 256, 55, dtype=np.uint8).tobytes())
        v = (2.0 * code - 1.0) * rng.uniform(1.0, 8.0, 1024)
        idx = rng.choice(1024, int(rng.integers(1, 40)), replace=False)
        v[idx] *= -rng.uniform(0.01, 0.5, idx.size)
        rows.append(v.astype(np.float32)); kind.append(4)
    return np.stack(rows), np.array(kind, np.uint8)


def _bulk_worker(args):
    lo, hi, L = args
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    import builtins
    llrs, _ = bulk_inputs(np)
    captured = {}

    def spy_sorted(seq, key=None):
        out = builtins.sorted(seq, key=key)
        if seq and hasattr(seq[0], "metric"):
            captured["metric"] = np.array([p.metric for p in out], dtype=np.float64)
            captured["u"] = np.stack([p.u.copy() for p in out])
        return out
    fp.sorted = spy_sorted
    pc = fp.PolarCode(1024, 448, list_size=L, crc_size=8)
    res = []
    for i in range(lo, hi):
        captured.clear()
        bits, ok = pc.decode(llrs[i])
        if captured:
            data = captured["u"][:, pc._data_pos]
            ci = np.packbits(data[:, :440], axis=1)
            cc = np.array([pc._crc_ok(d[:440], d[440:448]) for d in data], dtype=np.uint8)
            res.append((np.packbits(bits), bool(ok), True, ci, captured["metric"].copy(), cc))
        else:
            res.append((np.packbits(bits), bool(ok), False, None, None, None))
    return lo, res


def gen_bulk(mode, workers=4, L=8):
    import numpy as np
    import multiprocessing as mp
    llrs, kind = bulk_inputs(np)
    n = llrs.shape[0]
    step = 16
    jobs = [(lo, min(n, lo + step), L) for lo in range(0, n, step)]
    info = np.zeros((n, 55), np.uint8); ok = np.zeros(n, bool); took = np.zeros(n, bool)
    ci = np.zeros((n, L, 55), np.uint8); cm = np.zeros((n, L)); cc = np.zeros((n, L), np.uint8)
    with mp.get_context("spawn").Pool(workers) as pool:
        for lo, res in pool.imap_unordered(_bulk_worker, jobs):
            for j, r in enumerate(res):
                i = lo + j
                info[i], ok[i], took[i] = r[0], r[1], r[2]
                if r[2]:
                    ci[i], cm[i], cc[i] = r[3], r[4], r[5]
            print(f"  bulk[{mode}] {lo}..{lo + len(res)} done; ok {int(ok[lo:lo + len(res)].sum())} list {int(took[lo:lo + len(res)].sum())}", flush=True)
    out = {"info": info, "ok": ok, "took_list": took, "cand_info": ci, "cand_metric": cm, "cand_crc": cc,
           "meta_mode": np.array(mode), "meta_numpy": np.array(np.__version__), "list_size": np.array(L)}
    if mode == "glibc":
        out["llr"] = llrs; out["kind"] = kind
    np.savez_compressed(os.path.join(GOLD, f"polar_bulk_{mode}.npz"), **out)


# ----------------------------------------------------------------------------------------------- validator cases
def validator_inputs(np):
    """(llr float32[1024], list size, validator spec, ctr) cases.  Specs: ("reject",) ("call", k) ("raise",)
    ("payload", bytes) ("aead", ctr_expected)."""
    from echoseal_amd.polar_fast import encode
    from echoseal_amd.crypto import SecureChannel
    from echoseal_amd.embedder import synthetic_payloads
    rng = np.random.default_rng(515)
    sec = SecureChannel(KEY)
    cases = []
    ctrs = list(range(100, 100 + 96))
    sealed = synthetic_payloads(sec, ctrs, seed=99)
    for i, (c, blob) in enumerate(zip(ctrs, sealed)):
        code = encode(blob)
        L = (8, 8, 4, 32, 8, 16)[i % 6]
        family = i % 4
        if family == 0:                      # clean-ish: hard decision passes CRC
            v = (2.0 * code - 1.0) * rng.uniform(1.5, 6.0, 1024)
        elif family == 1:                    # a few weak flips: hard fails, the list may or may not recover
            v = (2.0 * code - 1.0) * rng.uniform(1.5, 6.0, 1024)
            idx = rng.choice(1024, int(rng.integers(1, 6)), replace=False); v[idx] *= -rng.uniform(0.02, 0.3, idx.size)
        elif family == 2:                    # AWGN sigma 0.3 .. 0.5
            s = rng.uniform(0.3, 0.5); v = np.clip(2.0 * (2.0 * code - 1.0 + rng.normal(0, s, 1024)) / s ** 2, -12, 12)
        else:                                # flips on the FIRST information positions (the unreliable ones)
            v = (2.0 * code - 1.0) * rng.uniform(2.0, 4.0, 1024)
            idx = rng.choice(64, int(rng.integers(1, 4)), replace=False); v[idx] *= -0.1
        v = v.astype(np.float32)
        specs = [("aead", c), ("aead", c + 1), ("reject",), ("call", 1 + i % 3), ("raise",), ("payload", blob)]
        for sp in (specs[i % 6], specs[(i + 1) % 6], specs[(i + 3) % 6]):
            cases.append((v, L, sp, c))
    return cases


def _validator_worker(args):
    lo, hi = args
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    from rtwm.crypto import SecureChannel as RefSec
    cases = validator_inputs(np)
    sec = RefSec(KEY)
    res = []
    for i in range(lo, hi):
        v, L, spec, c = cases[i]
        seen = []

        def validator(payload, spec=spec):
            seen.append(bytes(payload))
            if spec[0] == "reject":
                return False
            if spec[0] == "call":
                return len(seen) == spec[1]
            if spec[0] == "raise":
                raise RuntimeError("validator failure")
            if spec[0] == "payload":
                return bytes(payload) == spec[1]
            # the detector's closure, rtwm/detector.py:168-176, with the reference's SecureChannel
            try:
                pt = sec.open(payload)
            except Exception:
                return False
            if not pt.startswith(b"ESAL"):
                return False
            return int.from_bytes(pt[4:8], "big") == spec[1]
        pc = fp.PolarCode(1024, 448, list_size=L, crc_size=8)
        bits, ok = pc.decode(v, validator=validator)
        res.append((np.packbits(bits), bool(ok), seen))
    return lo, res


def gen_validator(workers=8):
    import numpy as np
    import multiprocessing as mp
    cases = validator_inputs(np)
    n = len(cases)
    jobs = [(lo, min(n, lo + 6)) for lo in range(0, n, 6)]
    out = {"count": np.array(n)}
    with mp.get_context("spawn").Pool(workers) as pool:
        for lo, res in pool.imap_unordered(_validator_worker, jobs):
            for j, (info, ok, seen) in enumerate(res):
                i = lo + j
                v, L, spec, c = cases[i]
                out[f"{i:03d}/llr"] = v; out[f"{i:03d}/L"] = np.array(L); out[f"{i:03d}/ctr"] = np.array(c)
                out[f"{i:03d}/spec"] = np.array(spec[0]); out[f"{i:03d}/arg"] = (
                    np.frombuffer(spec[1], np.uint8) if spec[0] == "payload" else np.array(spec[1] if len(spec) > 1 else -1))
                out[f"{i:03d}/info"] = info; out[f"{i:03d}/ok"] = np.array(ok)
                out[f"{i:03d}/seen"] = np.frombuffer(b"".join(seen), np.uint8).reshape(-1, 55) if seen else np.zeros((0, 55), np.uint8)
            print(f"  validator {lo}..{lo + len(res)}: ok {[r[1] for r in res]} calls {[len(r[2]) for r in res]}", flush=True)
    np.savez_compressed(os.path.join(GOLD, "polar_validator.npz"), **out)


# ----------------------------------------------------------------------------------------------- list sizes that are not powers of two
ODD_LISTS = (3, 5, 6, 12, 24, 100)


def gen_odd(mode):
    """PolarCode.decode with list sizes that are not powers of two (the reference takes any list_size >= 1)."""
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    from oracle.refshim import gen_golden as G
    base = np.load(os.path.join(GOLD, "polar_default.npz"))
    bulk = np.load(os.path.join(GOLD, "polar_bulk_glibc.npz"))
    cases = {"garbage_rng7": base["garbage_rng7/llr"], "det_clean_ctr0": base["det_clean_ctr0/llr"], "awgn070": base["awgn070/llr"],
             "c3_llr_17": bulk["llr"][17], "tie_heavy_600": bulk["llr"][600]}
    G.LISTS = ODD_LISTS
    out = G.run_polar(np, fp, cases)
    out["meta/mode"] = np.array(mode)
    np.savez_compressed(os.path.join(GOLD, f"polar_odd_{mode}.npz"), **out)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    if what == "all":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", "c3"], cwd=ROOT, env=env)
        for w in ("multi", "verify", "validator", "bulk", "odd"):
            subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", w], cwd=ROOT, env=env)
    elif what == "c3":
        gen_c3()
    elif what == "multi":
        gen_multi()
    elif what == "verify":
        gen_verify()
    elif what == "validator":
        gen_validator()
    elif what == "bulk":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", "bulk_default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", "bulk_glibc"], cwd=ROOT, env=env)
    elif what == "odd":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", "odd_default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r2", "odd_glibc"], cwd=ROOT, env=env)
    elif what in ("odd_default", "odd_glibc"):
        gen_odd(what.split("_")[1])
    elif what in ("bulk_default", "bulk_glibc"):
        gen_bulk(what.split("_")[1], workers=int(os.environ.get("GOLD_WORKERS", "6")))
    else:
        raise SystemExit(f"unknown target {what}")


if __name__ == "__main__":
    main()
