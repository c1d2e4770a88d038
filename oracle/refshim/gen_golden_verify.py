#!/usr/bin/env python3
"""Golden trace of the reference's full verify() search (build container only).

    python -m oracle.refshim.gen_golden_verify     # writes tests/golden/verify_trace.npz

Runs the reference WatermarkDetector.verify() / verify_raw_frame() on short synthetic clips with
list_size=1 (one decode ~0.1 s in the reference) and records, from the reference's own stdout,
the order in which (band, peak, counter) candidates were tried, plus the boolean result.
"""
import contextlib
import io
import os
import re
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(log: str):
    """-> list of (band_lo, peak_start, ctr) in the order tried."""
    out, band, peak = [], -1, -1
    for line in log.splitlines():
        m = re.match(r"\[SCAN\] Band \((\d+), (\d+)\)", line)
        if m:
            band = int(m.group(1))
        m = re.match(r"\s+Peak@(\d+):", line)
        if m:
            peak = int(m.group(1))
        m = re.match(r"\s+Trying ctr=(\d+)", line)
        if m:
            out.append((band, peak, int(m.group(1))))
    return np.array(out, dtype=np.int64).reshape(-1, 3)


def main():
    from oracle.refshim.shim import load_reference
    load_reference()
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.detector import WatermarkDetector
    key = b"\xAA" * 32
    sink = io.StringIO()
    out = {}
    rng = np.random.default_rng(4242)
    with contextlib.redirect_stdout(sink):
        tx = WatermarkEmbedder(key)
        tx._session_nonce = bytes(range(8))
        payloads = {}

        def fixed_payload(self):
            p = rng.integers(0, 256, 55, dtype=np.uint8).tobytes()
            payloads[self.frame_ctr] = p
            return p
        tx._build_payload = types.MethodType(fixed_payload, tx)
        clip = tx.process(np.zeros(12000, dtype=np.float32))          # 0.25 s of watermark over silence
    out["clip"] = clip.astype(np.float32)
    out["clip_payload_ctrs"] = np.array(sorted(payloads), dtype=np.int64)
    out["clip_payloads"] = np.stack([np.frombuffer(payloads[c], np.uint8) for c in sorted(payloads)])
    sink.seek(0); sink.truncate()
    with contextlib.redirect_stdout(sink):
        rx = WatermarkDetector(key, list_size=1)
        res = rx.verify(clip, 48000)
    out["clip_result"] = np.array(res)
    out["clip_trace"] = parse(sink.getvalue())
    print("verify(clip) ->", res, "tries:", out["clip_trace"].shape[0], file=sys.stderr)
    # single raw frame
    sink.seek(0); sink.truncate()
    with contextlib.redirect_stdout(sink):
        tx2 = WatermarkEmbedder(key)
        tx2._build_payload = types.MethodType(lambda s: bytes(range(55)), tx2)
        frame = tx2._make_frame_chips()
        rx2 = WatermarkDetector(key, list_size=1)
        sink.seek(0); sink.truncate()
        res2 = rx2.verify_raw_frame(frame)
    out["frame"] = frame
    out["frame_result"] = np.array(res2)
    out["frame_trace"] = parse(sink.getvalue())
    print("verify_raw_frame ->", res2, "scan tries:", out["frame_trace"].shape[0], file=sys.stderr)
    # degenerate inputs
    with contextlib.redirect_stdout(sink):
        out["empty_result"] = np.array(WatermarkDetector(key, list_size=1).verify(np.zeros(0, np.float32), 48000))
        out["short_result"] = np.array(WatermarkDetector(key, list_size=1).verify(np.zeros(40, np.float32), 48000))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "verify_trace.npz"), **out)


if __name__ == "__main__":
    main()
