"""Randomised parity of the wide list decoder (L = 64, 128, 256) against the oracle on tie-heavy (quantised) LLRs."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.oracle as orc
from echoseal_amd.engine import RxEngine
orc.build()
eng = RxEngine(0, list_size_max=256); dev = eng.device
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng(7000 + seed)
    B = 12
    q = rng.choice([-12.0, -6.0, -3.0, 0.0, 3.0, 6.0, 12.0], size=(B, 1024), p=[.1, .15, .2, .1, .2, .15, .1]).astype(np.float32)
    q[: B // 2] += rng.normal(0, 0.3, (B // 2, 1024)).astype(np.float32)
    x = torch.from_numpy(q).to(dev)
    for L in (64, 128, 256):
        a = eng.scl(x, list_size=L, skip_if_hard_ok=False)
        for i in range(B):
            nn, ci, cm, cc = orc.scl_list(q[i].astype(np.float64), L)
            ok = int(a.ncand[i]) == nn and np.array_equal(np.packbits(ci[:nn], axis=1), a.cand_info[i, :nn].cpu().numpy()) \
                and np.array_equal(cm[:nn], a.cand_metric[i, :nn].cpu().numpy()) and np.array_equal(cc[:nn], a.cand_ok[i, :nn].cpu().numpy())
            if not ok:
                bad += 1; print("wide vs oracle differ", seed, L, i)
    print(f"seed {seed}: mismatches = {bad}", flush=True)
print("FUZZ RESULT:", "clean" if bad == 0 else f"{bad} mismatches")
