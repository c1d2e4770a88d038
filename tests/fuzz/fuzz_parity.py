"""Randomised parity sweep (not part of the test suite: minutes of GPU time).  Inputs are chosen to provoke ties:
quantised LLRs for the list decoders (one-frame kernel vs 16-paths-per-wave kernel vs oracle), scaled / sparse /
repeated records for the float32 sync screen vs the float64 path, noise records for the screened shift search vs oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.oracle as orc
from echoseal_amd.engine import RxEngine
from echoseal_amd.tables import pack_tables
orc.build()
eng = RxEngine(0, list_size_max=32); dev = eng.device
ba, tpl, taps, ntaps, _ = pack_tables()
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
_KEY = b"\xAA" * 32; _tx = WatermarkEmbedder(_KEY); _c = list(range(32))
FR = _tx.make_frames(_c, synthetic_payloads(_tx.sec, _c)); FR_BAND = [band_index(_KEY, c) for c in _c]
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rng = np.random.default_rng(1000 + seed)
    # ---- list decoders on quantised LLRs (many exact ties in metrics and in f)
    B = 16 * 9 + 3
    q = rng.choice([-12.0, -6.0, -3.0, 0.0, 3.0, 6.0, 12.0], size=(B, 1024), p=[.1, .15, .2, .1, .2, .15, .1]).astype(np.float32)
    q[: B // 2] += rng.choice([0.0, 0.5], size=(B // 2, 1024)).astype(np.float32)
    x = torch.from_numpy(q).to(dev)
    for L in (1, 2, 4, 8, 16, 32, int(rng.choice([3, 5, 6, 7, 11, 13, 24, 29]))):
        eng.set_option("scl_multi", 0); a = eng.scl(x, list_size=L, skip_if_hard_ok=False)
        for lanes in (4, 2, 1):                      # the mappings of the several-frames-per-wave kernels (1: one lane per path, es_scl_wide.hip) against the one-frame kernel
            eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", lanes); b = eng.scl(x, list_size=L, skip_if_hard_ok=False)
            eng.set_option("scl_multi", -1); eng.set_option("scl_lanes", 0)
            for nm in ("hard_info", "hard_ok", "ncand", "cand_info", "cand_metric", "cand_ok"):
                if not torch.equal(getattr(a, nm), getattr(b, nm)):
                    bad += 1; print("SCL kernels differ", seed, L, lanes, nm)
        for i in rng.integers(0, B, 3):
            nn, ci, cm, cc = orc.scl_list(q[i].astype(np.float64), L)
            if not (np.array_equal(np.packbits(ci[:nn], axis=1), a.cand_info[i, :nn].cpu().numpy()) and np.array_equal(cm[:nn], a.cand_metric[i, :nn].cpu().numpy())):
                bad += 1; print("SCL vs oracle differ", seed, L, int(i))
    # ---- sync: float32 screen + exact picking vs float64 path
    T = int(rng.choice([1215, 1500, 2048, 3000]))
    n = 96
    r = rng.normal(0, 0.2, (n, T)).astype(np.float32)
    r[::5] *= (10.0 ** rng.uniform(-8, 8, (len(r[::5]), 1))).astype(np.float32)
    r[1::7, : T // 2] = 0.0
    r[2::11] = np.round(r[2::11] * 4) / 4
    r[3::13] = np.tile(r[3::13, :97], (1, T // 97 + 1))[:, :T]
    band = rng.integers(0, 4, n).astype(np.uint8)
    # planted frames of random strength (peaks anywhere between the noise floor and the 0.95 cap; sometimes two of them,
    # closer than the 607-lag exclusion zone or further apart)
    for i in range(4, n, 3):
        band[i] = FR_BAND[i % len(FR_BAND)]
        amp = 10.0 ** rng.uniform(-1.5, 0.7)
        for rep in range(int(rng.integers(1, 3))):
            off = int(rng.integers(0, T - 1215 + 1))
            r[i, off:off + 1215] += (amp * FR[i % len(FR)]).astype(np.float32)
    f = torch.from_numpy(r).to(dev); bb = torch.from_numpy(band).to(dev)
    ref = eng.sync(f, bb, keep_corr=False)
    k = (ref.npeaks & 0xFFFF).clamp(max=32); mask = torch.arange(32, device=dev)[None, :] < k[:, None]
    for fused in (True, False):                      # one kernel with the row in LDS / screen through HBM
        fast = eng.sync_fast(f, bb, fused=fused)
        if not (torch.equal(ref.thr, fast.thr) and torch.equal(ref.npeaks, fast.npeaks) and torch.equal(ref.peaks[mask], fast.peaks[mask])):
            bad += 1; print("sync fast vs float64 differ", seed, T, "fused" if fused else "unfused")
    # ---- LLR with the screened shift search vs oracle
    m = 24
    yy = rng.normal(0, rng.choice([1e-3, 0.1, 3.0]), (m, 1215))
    if seed % 3 == 0: yy[::2] = np.round(yy[::2] * 8) / 8
    bnd = rng.integers(0, 4, m).astype(np.uint8); pn = rng.integers(0, 256, (m, 152), dtype=np.uint8)
    llr, bs, sc = eng.llr(torch.from_numpy(yy).to(dev), torch.from_numpy(bnd).to(dev), torch.from_numpy(pn).to(dev), want_diag=True)
    llr = llr.cpu().numpy(); bs = bs.cpu().numpy(); sc = sc.cpu().numpy()
    for i in range(m):
        o, obs, best, second = orc.llr(yy[i], np.unpackbits(pn[i])[191:1215], taps[bnd[i], :ntaps[bnd[i]]])
        if not (obs == int(bs[i]) and np.array_equal(o, llr[i]) and np.float32(best) == sc[i, 0] and np.float32(second) == sc[i, 1]):
            bad += 1; print("LLR vs oracle differ", seed, i, obs, int(bs[i]))
    print(f"seed {seed}: ok so far, mismatches = {bad}", flush=True)
print("FUZZ RESULT:", "clean" if bad == 0 else f"{bad} mismatches")
