"""Every instantiation of the one-lane-per-path list decoder (LF = 1 .. 256) against the oracle, list sizes at and below the kernel's capacity,
on random rows, an all-zero row and tie-heavy rows; prints WHERE a mismatch is (which output, first row / byte).
    python tests/fuzz/wide_lf_check.py [BUILD_NAME]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
import oracle.oracle as orc
from echoseal_amd.engine import RxEngine
orc.build()
eng = RxEngine(0, list_size_max=256); dev = eng.device
rng = np.random.default_rng(123)
x = np.clip(rng.normal(0, 3, (12, 1024)), -12, 12).astype(np.float32)
x[0] = 0.0; x[0, 0] = 1e-3
x[1] = 0.0
x[2] = np.round(x[2]); x[3] = np.clip(rng.normal(0, 0.5, 1024), -12, 12)
xt = torch.from_numpy(x).to(dev)
eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", 1)
bad = 0
for L in (1, 2, 3, 4, 5, 8, 11, 16, 20, 32, 33, 48, 64, 100, 128, 200, 256):
    for skip in (False, True):
        r = eng.scl(xt, list_size=L, skip_if_hard_ok=skip); torch.cuda.synchronize()
        for i in range(x.shape[0]):
            nn, ci, cm, cc = orc.scl_list(x[i].astype(np.float64), L)
            info, hok = orc.polar_hard(x[i].astype(np.float64))
            if skip and hok:
                if int(r.ncand[i]) != 0: bad += 1; print(f"L={L} skip row {i}: ncand {int(r.ncand[i])} for a settled row")
                continue
            gi = r.cand_info[i].cpu().numpy(); gm = r.cand_metric[i].cpu().numpy(); gc = r.cand_ok[i].cpu().numpy()
            wi = np.packbits(ci, axis=1)
            msg = []
            if int(r.ncand[i]) != nn: msg.append(f"ncand {int(r.ncand[i])} != {nn}")
            if not np.array_equal(gm, cm): k = int(np.flatnonzero(gm != cm)[0]); msg.append(f"metric first diff at rank {k}: {gm[k]!r} vs {cm[k]!r} ({int((gm != cm).sum())} of {L})")
            if not np.array_equal(gi, wi): k = int(np.flatnonzero((gi != wi).any(1))[0]); b = int(np.flatnonzero(gi[k] != wi[k])[0]); msg.append(f"info first diff at rank {k} byte {b} ({int((gi != wi).any(1).sum())} rows)")
            if not np.array_equal(gc, cc): msg.append("crc flags")
            if msg: bad += 1; print(f"L={L} skip={int(skip)} row {i}: " + "; ".join(msg), flush=True)
    print(f"L={L}: done, mismatching rows so far {bad}", flush=True)
print("WIDE LF CHECK:", "clean" if bad == 0 else f"{bad} mismatching rows")
