"""Repeat-run determinism of the lane-per-path list decoder at L = 64 / 128 / 256 (several waves per frame: any missing barrier would
show up as run-to-run differences), plus a spot check of a few frames against the oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.oracle as orc
from echoseal_amd.engine import RxEngine
orc.build()
eng = RxEngine(0, list_size_max=256); rng = np.random.default_rng(5)
B = 3072
q = np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)
q[::5] = rng.choice([-12.0, -6.0, -3.0, 0.0, 3.0, 6.0, 12.0], size=(len(q[::5]), 1024)).astype(np.float32)      # tie-heavy rows
x = torch.from_numpy(q).to(eng.device)
bad = 0
for L in (64, 128, 256):
    ref = eng.scl(x, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
    for rep in range(4):
        r = eng.scl(x, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        for nm in ("cand_info", "cand_metric", "cand_ok", "ncand"):
            if not torch.equal(getattr(ref, nm), getattr(r, nm)):
                bad += 1; print("run-to-run difference", L, rep, nm)
    for i in rng.integers(0, B, 4):
        nn, ci, cm, cc = orc.scl_list(q[i].astype(np.float64), L)
        if not (np.array_equal(np.packbits(ci[:nn], axis=1), ref.cand_info[i, :nn].cpu().numpy()) and np.array_equal(cm[:nn], ref.cand_metric[i, :nn].cpu().numpy())):
            bad += 1; print("differs from the oracle", L, int(i))
    print(f"L={L}: 5 runs of {B} frames compared, 4 frames against the oracle; problems so far {bad}", flush=True)
# ---- frames drawn from the launch's counter (skip_if_hard_ok): which wave decodes a frame varies from run to run, its rows must not
info = torch.from_numpy(rng.integers(0, 256, (B, 55), dtype=np.uint8)).to(eng.device)
code = eng.polar_encode(info).cpu().numpy().astype(np.float64)
clean = (2.0 * code - 1.0) * 6.0
mix = np.where((rng.random(B) < 0.6)[:, None], q, clean).astype(np.float32)            # 40 % of the frames pass the hard decision
xm = torch.from_numpy(mix).to(eng.device)
eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", 1)
for L in (2, 8, 16, 64):
    full = eng.scl(xm, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
    took = ~full.hard_ok.bool()
    for rep in range(6):
        r = eng.scl(xm, list_size=L, skip_if_hard_ok=True); torch.cuda.synchronize()
        for nm in ("cand_info", "cand_metric", "cand_ok", "ncand"):
            if not torch.equal(getattr(full, nm)[took], getattr(r, nm)[took]) or bool(getattr(r, nm)[~took].to(torch.float64).abs().sum().item()):
                bad += 1; print("drawn frames: difference", L, rep, nm)
        if not (torch.equal(full.hard_info, r.hard_info) and torch.equal(full.hard_ok, r.hard_ok)):
            bad += 1; print("drawn frames: hard decision differs", L, rep)
    print(f"L={L}: 6 launches with drawn frames ({int(took.sum())} of {B} listed) equal the fixed-group launch; problems so far {bad}", flush=True)
print("RESULT:", "clean" if bad == 0 else f"{bad} problems")
