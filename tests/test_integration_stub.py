"""The reference-side ctypes stub printed in INTEGRATION.md section 2 (`rtwm/_hip.py`) is executed VERBATIM here, so that it cannot
drift from include/echoseal_hip.h again (round 3: the stub still said float32[4,160] after the tap table's row stride had become 576)."""
import os
import re
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source() -> str:
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# rtwm/_hip\.py.*?)```", md, re.S)
    assert m, "INTEGRATION.md no longer holds the rtwm/_hip.py block"
    return m.group(1)


def test_stub_states_the_header_constants():
    """CPU: the constants the stub hard-codes are the header's; it compiles; it binds only exported names."""
    src = stub_source()
    hdr = open(os.path.join(ROOT, "include", "echoseal_hip.h")).read()
    compile(src, "INTEGRATION.md:rtwm/_hip.py", "exec")
    names = re.search(r"ES_ABI_VERSION, ES_MAX_TAPS, ES_MAX_PEAKS, ES_PN_BYTES = (\d+), (\d+), (\d+), (\d+)", src)
    for const, val in zip(("ES_ABI_VERSION", "ES_MAX_TAPS", "ES_MAX_PEAKS", "ES_PN_BYTES"), names.groups()):
        assert int(re.search(rf"#define\s+{const}\s+(\d+)", hdr).group(1)) == int(val), const
    declared = set(re.findall(r"\b(es_[a-z0-9_]+)\s*\(", hdr))
    assert set(re.findall(r"\b(es_[a-z0-9_]+)\b", src)) - {"es_ctx"} <= declared
    assert "float32[4,160]" not in open(os.path.join(ROOT, "INTEGRATION.md")).read()


@pytest.mark.gpu
def test_stub_runs_verbatim_and_decodes_frames(engine, oracle, monkeypatch):
    import torch
    import echoseal_amd._native as nat
    from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
    from echoseal_amd.tables import pack_tables
    from echoseal_amd.utils import band_index
    monkeypatch.setenv("ECHOSEAL_HIP_LIB", nat.LIB_PATH)
    hip = types.ModuleType("rtwm_hip_stub")
    exec(compile(stub_source(), "INTEGRATION.md:rtwm/_hip.py", "exec"), hip.__dict__)
    try:
        key = b"\xAA" * 32
        tx = WatermarkEmbedder(key)
        ctrs = list(range(8))
        frames = tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))
        frames[4:] = (frames[4:] + np.random.default_rng(11).normal(0, 0.2, frames[4:].shape)).astype(np.float32)
        band = np.array([band_index(key, c) for c in ctrs], np.uint8)
        pn = tx.sec.pn_bytes_batch(ctrs, 152)
        ba, tpl, taps, ntaps, frozen = pack_tables()
        hip.set_tables(ba, tpl, taps, ntaps, frozen)
        with pytest.raises(AssertionError):
            hip.set_tables(ba, tpl, np.ascontiguousarray(taps[:, :160]), ntaps, frozen)        # the old stride is refused, not read out of bounds
        d = torch.device("cuda:0")
        y, thr, peaks, npk, llr = hip.front(torch.from_numpy(frames).to(d), torch.from_numpy(band).to(d), torch.from_numpy(pn).to(d))
        hard, hok, ci, cm, co, nc = hip.scl(llr, 8)
        torch.cuda.synchronize()
        for i in range(8):
            o = oracle.decode_frame(frames[i], ba[band[i]], tpl[band[i]], taps[band[i], :ntaps[band[i]]], np.unpackbits(pn[i])[:1215], L=8)
            n = int(npk[i])
            assert list(peaks[i, :n].cpu().numpy()) == list(o["peaks"][:n]) and float(thr[i]) == o["thr"]
            assert np.array_equal(o["llr"], llr[i].cpu().numpy())
            info, ok = oracle.polar_hard(o["llr"].astype(np.float64))
            assert bool(hok[i]) == ok and np.array_equal(np.packbits(info), hard[i].cpu().numpy())
            if not ok:
                nn, oi, om, oc = oracle.scl_list(o["llr"].astype(np.float64), 8)
                assert int(nc[i]) == nn == 8
                assert np.array_equal(np.packbits(oi, axis=1), ci[i].cpu().numpy()) and np.array_equal(om, cm[i].cpu().numpy())
                assert np.array_equal(oc, co[i].cpu().numpy())
    finally:
        hip._lib.es_destroy.argtypes = [__import__("ctypes").c_void_p]
        hip._lib.es_destroy(hip._ctx)
