#!/usr/bin/env python3
"""Time the REFERENCE itself on this container's host cores (build container only; SURVEY section 8d asks for the
reference's own NumPy path beside the GPU numbers).  One metric unit = sync (band-pass, NCC, threshold, peak pick)
+ _llr(variant 0) + polar decode with list size 8, validator None, on clean frames from the reference's embedder.

    python -m oracle.refshim.time_reference [n_frames]      # writes profiles/r01_reference_cpu_timing.json

The reference is imported through oracle/refshim/shim.py and RUN; nothing of it is stored."""
from __future__ import annotations

import contextlib, io, json, os, platform, sys, time, types

import numpy as np

from .shim import load_reference

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    load_reference()
    from scipy.signal import lfilter, correlate
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.detector import WatermarkDetector
    from rtwm.utils import choose_band, butter_bandpass
    from rtwm import polar_fast
    key = b"\xAA" * 32
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        tx = WatermarkEmbedder(key); rx = WatermarkDetector(key, list_size=8)
    t_sync, t_llr, t_scl = [], [], []
    for ctr in range(n):
        with contextlib.redirect_stdout(sink):
            tx.frame_ctr = ctr
            frame = tx._make_frame_chips()
            band = choose_band(key, ctr)
            t0 = time.perf_counter()
            b, a = butter_bandpass(*band, 48000, order=4)
            y = lfilter(b, a, frame.astype(np.float32, copy=False))
            tpl = lfilter(b, a, lfilter(b, a, rx._pre_sy)); tpl = tpl / float(np.sqrt(np.sum(tpl * tpl)) + 1e-12)
            e_y = np.sqrt(np.convolve(y * y, np.ones(63, dtype=np.float32), mode="valid")) + 1e-12
            corr = correlate(y, tpl, mode="valid") / e_y
            med = float(np.median(corr)); mad = float(np.median(np.abs(corr - med))) + 1e-12
            thr = min(med + 4.5 * 1.4826 * mad, 0.95)
            peaks = [i for i in range(corr.size) if corr[i] >= thr and corr[i] >= np.max(corr[max(0, i - 607):i + 608])]
            t1 = time.perf_counter()
            llr = rx._llr(y, ctr, 0)
            t2 = time.perf_counter()
            polar_fast.decode(llr, list_size=8, return_ok=True)
            t3 = time.perf_counter()
        t_sync.append(t1 - t0); t_llr.append(t2 - t1); t_scl.append(t3 - t2)
    tot = np.array(t_sync) + np.array(t_llr) + np.array(t_scl)
    out = {
        "what": "the reference (PetarSt98/EchoSeal @ /root/reference) run in the build container, single process, one frame at a time",
        "unit": "sync (rtwm/detector.py:59-99 as inlined in oracle/refshim/gen_golden.py) + WatermarkDetector._llr(variant 0) + "
                "polar_fast.decode(list_size=8, validator=None)",
        "frames": n, "host": platform.processor() or platform.machine(), "cpus_visible": os.cpu_count(),
        "python": platform.python_version(), "numpy": np.__version__,
        "median_s": {"sync": float(np.median(t_sync)), "llr": float(np.median(t_llr)), "scl8": float(np.median(t_scl)),
                     "total": float(np.median(tot))},
        "frames_per_s_per_core": float(1.0 / np.median(tot)),
    }
    path = os.path.join(ROOT, "profiles", "r01_reference_cpu_timing.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
