#!/usr/bin/env python3
"""Round-3 golden vectors: BASELINE config 5's list sweep (L = 1 / 4 / 16; L = 8 is polar_bulk_*.npz) pinned to the
REFERENCE, captured by running it (build container only; /root/reference is read-only and never travels).  Only inputs
and outputs are stored under tests/golden/ -- no reference source.

    python -m oracle.refshim.gen_golden_r3 [all|inputs|sweep_default|sweep_glibc|codes|codes2|fs]

    polar_sweep_{default,glibc}.npz   for each L in (1, 4, 16): 256 LLR vectors through the reference's
        PolarCode.decode(list_size=L) (rtwm/fastpolar.py:254-359): (info, ok), whether the list loop ran, and the final
        list in the order of the reference's last sort (bits, metrics, CRC flags).  Mix per L (disjoint between the Ls):
          64  detector-produced: the reference's _llr (PN variant 1) on BASELINE-config-3 windows
              (c3_windows.npz rows, computed by the reference in round 2: rtwm/detector.py:296-416)
          64  detector-produced through the config-5 SURROGATE channel (echoseal_amd.workloads.lossy_channel, NOT MP3):
              reference embedder frame -> lossy_channel -> reference band-pass (scipy lfilter) -> reference _llr
              (variant 0, start 0)
          64  AWGN on random codewords, sigma 0.3 .. 1.1, clipped to +-12
          64  tie-heavy (few distinct magnitudes, zeros, +-12)
        Both NumPy run-time modes (see gen_golden.py: `default` = AVX-512 exp/log1p dispatch, `glibc` = C library).

    polar_codes_{default,glibc}.npz   PolarCode(1024, K, list_size=L, crc_size=8).decode for K in (16, 64, 200, 512, 1000) -- the
        reference's PolarCode takes any K (rtwm/fastpolar.py:209-234); byte-aligned K is what rtwm/polar_fast.py can round-trip -- and
        L in (1, 8): six LLR vectors per K (codeword in light / heavy AWGN, garbage, +-12 ties on a codeword with flips, a clean
        codeword at +-3, zeros), (info, ok), whether the list loop ran, and the final list.
    polar_codes2_{default,glibc}.npz  the same for K in (9, 13, 301, 1023, 1024): one information bit, K - 8 not a whole number of bytes
        (rows are np.packbits of the information bits: zero padding), and a code without any frozen position.
"""
from __future__ import annotations

import contextlib
import io
import os
import subprocess
import sys

AVX512_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
KEY = b"\xAA" * 32
LISTS = (1, 4, 16)
PER_L = 256
INPUTS = os.path.join(GOLD, "polar_sweep_inputs.npz")


def gen_inputs():
    """The 3 x 256 LLR vectors (float32).  The detector-produced rows come from the reference's own _llr."""
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    from rtwm.detector import WatermarkDetector
    from rtwm.utils import choose_band, butter_bandpass
    from scipy.signal import lfilter
    from echoseal_amd.polar_fast import encode
    from echoseal_amd.workloads import lossy_channel
    from oracle.refshim.gen_golden_r2 import ref_frames
    c3 = np.load(os.path.join(GOLD, "c3_windows.npz"))
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        rx = WatermarkDetector(KEY, list_size=8)
    ctrs = list(range(1000, 1000 + 64 * len(LISTS)))                  # frames no other fixture uses
    frames, _ = ref_frames(np, ctrs)
    lossy = lossy_channel(frames)
    out, kinds = {}, {}
    for li, L in enumerate(LISTS):
        rng = np.random.default_rng(30260 + L)
        rows, kind = [], []
        for i in range(64):                                           # reference _llr, PN variant 1, on C3 windows
            rows.append(c3["llr1"][64 * li + i].astype(np.float32)); kind.append(0)
        for i in range(64):                                           # reference _llr on the lossy-channel frames
            c = ctrs[64 * li + i]
            b, a = butter_bandpass(*choose_band(KEY, c), 48000, order=4)
            y = lfilter(b, a, lossy[64 * li + i].astype(np.float32, copy=False))
            with contextlib.redirect_stdout(sink):
                rows.append(np.asarray(rx._llr(y, c, 0), np.float32))
            kind.append(1)
        for i in range(64):                                           # AWGN on random codewords
            sigma = (0.3, 0.45, 0.6, 0.75, 0.9, 1.0, 1.1, 0.5)[i % 8]
            code = encode(rng.integers(0, 256, 55, dtype=np.uint8).tobytes())
            r = 2.0 * code.astype(np.float64) - 1.0 + rng.normal(0, sigma, 1024)
            rows.append(np.clip(2.0 * r / sigma ** 2, -12, 12).astype(np.float32)); kind.append(2)
        for i in range(64):                                           # tie-heavy
            levels = ((0.0, 12.0), (1.0,), (0.5, 1.0, 1.5), (0.0, 2.0, 4.0), (12.0,), (0.0, 0.25), (0.0,), (3.0, 12.0))[i % 8]
            mag = rng.choice(np.array(levels), 1024)
            sign = rng.integers(0, 2, 1024) * 2.0 - 1.0
            v = (mag * sign).astype(np.float32)
            if i % 16 >= 8 and levels[-1]:                            # partly a real codeword at one magnitude
                code = encode(rng.integers(0, 256, 55, dtype=np.uint8).tobytes())
                flips = rng.random(1024) < 0.04
                v = ((2.0 * (code ^ flips) - 1.0) * levels[-1]).astype(np.float32)
            rows.append(v); kind.append(3)
        out[f"L{L}/llr"] = np.stack(rows)
        kinds[f"L{L}/kind"] = np.array(kind, np.uint8)
        print(f"  inputs L={L}: {len(rows)} rows", flush=True)
    np.savez_compressed(INPUTS, **out, **kinds)


def _worker(args):
    L, lo, hi = args
    import builtins
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    llrs = np.load(INPUTS)[f"L{L}/llr"]
    captured = {}

    def spy_sorted(seq, key=None):
        out = builtins.sorted(seq, key=key)
        if seq and hasattr(seq[0], "metric"):
            captured["metric"] = np.array([p.metric for p in out], dtype=np.float64)
            captured["u"] = np.stack([p.u.copy() for p in out])
        return out
    fp.sorted = spy_sorted                                            # module-global lookup precedes builtins
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
    return L, lo, res


def gen_sweep(mode, workers=6):
    import multiprocessing as mp
    import numpy as np
    inp = np.load(INPUTS)
    out = {"meta_mode": np.array(mode), "meta_numpy": np.array(np.__version__), "lists": np.array(LISTS)}
    store = {}
    jobs = []
    for L in LISTS:
        n = inp[f"L{L}/llr"].shape[0]
        store[L] = dict(info=np.zeros((n, 55), np.uint8), ok=np.zeros(n, bool), took=np.zeros(n, bool),
                        ci=np.zeros((n, L, 55), np.uint8), cm=np.zeros((n, L)), cc=np.zeros((n, L), np.uint8), nc=np.zeros(n, np.int32))
        step = 32 if L == 1 else 16 if L == 4 else 8
        jobs += [(L, lo, min(n, lo + step)) for lo in range(0, n, step)]
    jobs.sort(key=lambda j: -j[0])                                   # the long ones first
    with mp.get_context("spawn").Pool(workers) as pool:
        for L, lo, res in pool.imap_unordered(_worker, jobs):
            s = store[L]
            for j, r in enumerate(res):
                i = lo + j
                s["info"][i], s["ok"][i], s["took"][i] = r[0], r[1], r[2]
                if r[2]:
                    k = r[4].shape[0]
                    s["nc"][i] = k
                    s["ci"][i, :k], s["cm"][i, :k], s["cc"][i, :k] = r[3], r[4], r[5]
            print(f"  sweep[{mode}] L={L} {lo}..{lo + len(res)}: ok {int(s['ok'][lo:lo + len(res)].sum())} list {int(s['took'][lo:lo + len(res)].sum())}", flush=True)
    for L in LISTS:
        s = store[L]
        out.update({f"L{L}/info": s["info"], f"L{L}/ok": s["ok"], f"L{L}/took_list": s["took"], f"L{L}/cand_info": s["ci"],
                    f"L{L}/cand_metric": s["cm"], f"L{L}/cand_crc": s["cc"], f"L{L}/ncand": s["nc"]})
        if mode == "glibc":
            out[f"L{L}/llr"] = inp[f"L{L}/llr"]; out[f"L{L}/kind"] = inp[f"L{L}/kind"]
    np.savez_compressed(os.path.join(GOLD, f"polar_sweep_{mode}.npz"), **out)


CODES_K = (16, 64, 200, 512, 1000)
CODES2_K = (9, 13, 301, 1023, 1024)        # K that is not a whole number of bytes; one information bit; no frozen position at all
CODES_L = (1, 8)


def gen_codes(mode, ks=CODES_K, name="polar_codes"):
    import builtins
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    captured = {}

    def spy_sorted(seq, key=None):
        out = builtins.sorted(seq, key=key)
        if seq and hasattr(seq[0], "metric"):
            captured["metric"] = np.array([p.metric for p in out], dtype=np.float64)
            captured["u"] = np.stack([p.u.copy() for p in out])
        return out
    fp.sorted = spy_sorted
    out = {"meta_mode": np.array(mode), "meta_numpy": np.array(np.__version__), "ks": np.array(ks), "lists": np.array(CODES_L)}
    for K in ks:
        rng = np.random.default_rng(4000 + K)
        enc = fp.PolarCode(1024, K, list_size=1, crc_size=8)
        rows = []
        for kind in range(6):
            code = enc.encode(rng.integers(0, 2, K - 8, dtype=np.uint8)).astype(np.float64)
            if kind == 0: v = np.clip(2.0 * (2.0 * code - 1.0 + rng.normal(0, 0.5, 1024)) / 0.25, -12, 12)
            elif kind == 1: v = np.clip(2.0 * (2.0 * code - 1.0 + rng.normal(0, 1.0, 1024)) / 1.0, -12, 12)
            elif kind == 2: v = np.clip(rng.normal(0, 3.0, 1024), -12, 12)
            elif kind == 3:
                flips = rng.random(1024) < 0.05
                v = (2.0 * np.logical_xor(code > 0, flips) - 1.0) * 12.0
            elif kind == 4: v = (2.0 * code - 1.0) * 3.0
            else: v = np.zeros(1024)
            rows.append(v.astype(np.float32))
        llrs = np.stack(rows)
        if mode == "glibc":
            out[f"K{K}/llr"] = llrs
        for L in CODES_L:
            pc = fp.PolarCode(1024, K, list_size=L, crc_size=8)
            nb = (K - 8 + 7) // 8
            n = llrs.shape[0]
            info = np.zeros((n, nb), np.uint8); ok = np.zeros(n, bool); took = np.zeros(n, bool)
            ci = np.zeros((n, L, nb), np.uint8); cm = np.zeros((n, L)); cc = np.zeros((n, L), np.uint8); nc = np.zeros(n, np.int32)
            for i in range(n):
                captured.clear()
                bits, okk = pc.decode(llrs[i])
                info[i] = np.packbits(bits); ok[i] = bool(okk)
                if captured:
                    took[i] = True
                    data = captured["u"][:, pc._data_pos]
                    k = data.shape[0]; nc[i] = k
                    ci[i, :k] = np.packbits(data[:, :K - 8], axis=1)
                    cm[i, :k] = captured["metric"]
                    cc[i, :k] = [pc._crc_ok(d[:K - 8], d[K - 8:K]) for d in data]
            t = f"K{K}/L{L}"
            out.update({f"{t}/info": info, f"{t}/ok": ok, f"{t}/took_list": took, f"{t}/cand_info": ci, f"{t}/cand_metric": cm,
                        f"{t}/cand_crc": cc, f"{t}/ncand": nc})
            print(f"  codes[{mode}] K={K} L={L}: ok {int(ok.sum())} list {int(took.sum())}", flush=True)
    np.savez_compressed(os.path.join(GOLD, f"{name}_{mode}.npz"), **out)


FS_TARGETS = (44_100, 96_000)


def gen_fs(n=24):
    """WatermarkDetector(fs_target = 44 100 / 96 000) (rtwm/detector.py:27): the band-pass design, the preamble template and the matched
    filter follow the rate -- 550 taps in the 18-22 kHz band at 44 100 Hz, 229..262 at 96 000 -- and with them the shift range of _llr and
    _decode_header.  The first `n` C3 windows (the samples are what they are: the detector is told another rate), everything
    _scan_band_multi_frame computes and _llr (both PN variants) at the first peak, as in gen_golden_r2.gen_c3."""
    import contextlib, io
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.detector as det_mod
    from rtwm.detector import WatermarkDetector
    from rtwm.utils import choose_band, BAND_PLAN, butter_bandpass
    from scipy.signal import lfilter
    from echoseal_amd.workloads import c3_windows
    from oracle.refshim.gen_golden_r2 import scan_capture, ref_frames, best_s_from
    ctrs = list(range(n))
    frames, payloads = ref_frames(np, ctrs)
    win, offs, facs, lens = c3_windows(frames)
    for fs in FS_TARGETS:
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            rx = WatermarkDetector(KEY, fs_target=fs, list_size=8)
        out = {"win": win[:n], "ctr": np.array(ctrs, np.int64), "fs": np.array(fs)}
        bands, thr, fb, npk, peaks, nvis, hdrs, llr0, llr1, bs, ntaps = [], [], [], [], [], [], [], [], [], [], []
        for i in range(n):
            band = choose_band(KEY, ctrs[i])
            cap = scan_capture(np, det_mod, rx, win[i], band)
            bands.append(BAND_PLAN.index(band))
            thr.append(cap["thr"]); fb.append(cap["fallback"])
            pk = np.full(32, -1, np.int32); k = min(32, cap["peaks"].size); pk[:k] = cap["peaks"][:k]
            peaks.append(pk); npk.append(cap["peaks"].size)
            h = np.zeros((5, 3)); v = cap["hdr"][:5]; h[:v.shape[0]] = v
            hdrs.append(h); nvis.append(cap["visited"].size)
            if i < 4:
                out[f"corr/{i:03d}"] = cap["corr"]
            start = int(cap["peaks"][0])
            b, a = butter_bandpass(*band, fs, order=4)
            y = lfilter(b, a, win[i].astype(np.float32, copy=False))
            with contextlib.redirect_stdout(sink):
                sink.seek(0); sink.truncate()
                l0 = rx._llr(y[start:start + 1215], ctrs[i], 0)
                l1 = rx._llr(y[start:start + 1215], ctrs[i], 1)
                ntaps.append(len(rx._matched_filter_taps(band)))
            b_s = best_s_from(sink.getvalue())
            llr0.append(l0); llr1.append(l1); bs.append((b_s + [0, 0])[:2] if len(b_s) >= 2 else [9999, 9999])
            print(f"  fs {fs} window {i}: band {bands[-1]} taps {ntaps[-1]} thr {cap['thr']:.4f} peaks {cap['peaks'][:3]} best_s {bs[-1]}", flush=True)
        out.update(band=np.array(bands, np.uint8), thr=np.array(thr), fallback=np.array(fb), npeaks=np.array(npk, np.int32), peaks=np.stack(peaks),
                   nvisited=np.array(nvis, np.int32), hdr=np.stack(hdrs), llr0=np.stack(llr0), llr1=np.stack(llr1), best_s=np.array(bs, np.int32),
                   ntaps=np.array(ntaps, np.int32))
        np.savez_compressed(os.path.join(GOLD, f"fs{fs}_windows.npz"), **out)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    if what == "all":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "inputs"], cwd=ROOT, env=env)
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "sweep_default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "sweep_glibc"], cwd=ROOT, env=env)
        os.remove(INPUTS)                                             # (the inputs live in polar_sweep_glibc.npz)
    elif what == "inputs":
        gen_inputs()
    elif what == "codes":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "codes_default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "codes_glibc"], cwd=ROOT, env=env)
    elif what in ("codes_default", "codes_glibc"):
        gen_codes(what.split("_")[1])
    elif what == "codes2":
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "codes2_default"], cwd=ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_r3", "codes2_glibc"], cwd=ROOT, env=env)
    elif what in ("codes2_default", "codes2_glibc"):
        gen_codes(what.split("_")[1], CODES2_K, "polar_codes2")
    elif what == "fs":
        gen_fs()
    elif what in ("sweep_default", "sweep_glibc"):
        gen_sweep(what.split("_")[1], workers=int(os.environ.get("GOLD_WORKERS", "6")))
    else:
        raise SystemExit(f"unknown target {what}")


if __name__ == "__main__":
    main()
