#!/usr/bin/env python3
"""BASELINE config 1 as the reference's own test writes it (tests/test_roundtrip_quick.py:5-15):
one synthesised frame through WatermarkDetector(key, list_size=32).verify_raw_frame (build container only).

    python -m oracle.refshim.gen_golden_quick32     # writes tests/golden/quick32.npz  (about 2-3 minutes of reference time)

Recorded from the reference run itself:
  * the frame (payload frozen: the embedder's draws from `secrets`), the boolean result, session_nonce afterwards;
  * every counter handed to _try_decode_frame, in order (the four direct tries of rtwm/detector.py:235-245, then the scan's),
    and the scan's (band, peak, counter) trace parsed from the reference's own stdout;
  * every polar decode the detector made (rtwm/detector.py:177-190): the LLR vector it passed, the list size, and the blob
    (or None) it got back with the detector's AEAD validator in place;
  * for each of those LLR vectors, PolarCode.decode(llr) WITHOUT a validator at list size 32: (info bits, ok) -- the list decode
    on the quick test's own LLRs, independent of the validator.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = 32


def main():
    from oracle.refshim.shim import load_reference
    from oracle.refshim.gen_golden_verify import parse
    load_reference()
    import rtwm.detector as RD
    from rtwm.embedder import WatermarkEmbedder
    from rtwm.fastpolar import PolarCode
    key = b"\xAA" * 32
    sink = io.StringIO()
    calls = []          # (llr float32[1024], list_size, blob or None)
    tried = []          # counters handed to _try_decode_frame
    real_dec = RD.polar_dec

    def recording_dec(llr, **kw):
        blob = real_dec(llr, **kw)
        calls.append((np.array(llr, dtype=np.float32, copy=True), int(kw.get("list_size", 8)), blob))
        return blob
    RD.polar_dec = recording_dec
    try:
        with contextlib.redirect_stdout(sink):
            tx = WatermarkEmbedder(key)
            tx._build_payload = types.MethodType(lambda s: bytes(range(55)), tx)
            frame = tx._make_frame_chips()
            rx = RD.WatermarkDetector(key, list_size=L)
            real_try = rx._try_decode_frame

            def recording_try(fr, ctr):
                tried.append(int(ctr))
                return real_try(fr, ctr)
            rx._try_decode_frame = recording_try
            sink.seek(0); sink.truncate()
            res = rx.verify_raw_frame(frame)
    finally:
        RD.polar_dec = real_dec
    print(f"verify_raw_frame(list_size={L}) -> {res}; {len(tried)} tries, {len(calls)} polar decodes", file=sys.stderr)
    out = {"frame": np.asarray(frame), "result": np.array(res), "list_size": np.array(L),
           "session_nonce": np.frombuffer(rx.session_nonce or b"", np.uint8),
           "tried": np.array(tried, np.int64), "scan_trace": parse(sink.getvalue()),
           "llr": np.stack([c[0] for c in calls]), "dec_list_size": np.array([c[1] for c in calls], np.int64),
           "blob_ok": np.array([c[2] is not None for c in calls]),
           "blob": np.stack([np.frombuffer(c[2], np.uint8) if c[2] is not None else np.zeros(55, np.uint8) for c in calls])}
    pc = PolarCode(1024, 448, list_size=L)
    info, ok = [], []
    for k, c in enumerate(calls):
        with contextlib.redirect_stdout(sink):
            b, o = pc.decode(c[0])
        info.append(np.asarray(b, np.uint8)); ok.append(bool(o))
        print(f"  plain decode {k + 1}/{len(calls)} ok={o}", file=sys.stderr)
    out["plain_info"] = np.packbits(np.stack(info), axis=1)
    out["plain_ok"] = np.array(ok)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "quick32.npz"), **out)


if __name__ == "__main__":
    main()
