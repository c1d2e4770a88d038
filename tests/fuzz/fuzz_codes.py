"""Randomised parity of the run-time-K list decoder (es_scl_wide_kernel<L, LF, true>) against the oracle built for the same code:
random K in 9..1024, list sizes 1..256 (powers of two and not), ragged batches, float32 / float64 LLRs of several kinds.
    python3 tests/fuzz/fuzz_codes.py [SEEDS]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.oracle as orc
from echoseal_amd.engine import RxEngine
orc.build()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    rng = np.random.default_rng(9100 + seed)
    K = int(rng.choice([9, 10, 15, 17, 31, 32, 33, 100, 255, 447, 449, 600, 777, 1001, 1023, 1024])) if seed % 2 else int(rng.integers(9, 1025))
    eng = RxEngine(0, list_size_max=256, code_k=K); dev = eng.device
    B = int(rng.integers(1, 30))
    kind = seed % 3
    if kind == 0:
        q = np.clip(rng.normal(0, 3, (B, 1024)), -12, 12)
    elif kind == 1:
        q = rng.choice([-12.0, -6.0, -3.0, 0.0, 3.0, 6.0, 12.0], size=(B, 1024), p=[.1, .15, .2, .1, .2, .15, .1])
        q[: B // 2] += rng.normal(0, 0.3, (B // 2, 1024))
    else:
        q = rng.normal(0, 40, (B, 1024))                       # unclipped: the softplus fall-back ranges
    dt = np.float32 if seed % 4 < 2 else np.float64
    q = q.astype(dt)
    x = torch.from_numpy(q).to(dev)
    with orc.code_k(K):
        for L in sorted({1, 2, int(rng.integers(3, 9)), 8, int(rng.integers(9, 33)), int(rng.integers(33, 65)), int(rng.integers(65, 257))}):
            a = eng.scl(x, list_size=L, skip_if_hard_ok=False).check()
            for i in range(B if L <= 32 else min(B, 6)):
                hinfo, hok = orc.polar_hard(q[i].astype(np.float64))
                nn, ci, cm, cc = orc.scl_list(q[i].astype(np.float64), L)
                ok = int(a.ncand[i]) == nn and np.array_equal(np.packbits(ci[:nn], axis=1), a.cand_info[i, :nn].cpu().numpy()) \
                    and np.array_equal(cm[:nn], a.cand_metric[i, :nn].cpu().numpy()) and np.array_equal(cc[:nn], a.cand_ok[i, :nn].cpu().numpy()) \
                    and np.packbits(hinfo).tobytes() == a.hard_info[i].cpu().numpy().tobytes() and hok == bool(a.hard_ok[i])
                if not ok:
                    bad += 1; print("differ: seed", seed, "K", K, "L", L, "row", i)
    eng.close()
    print(f"seed {seed}: K = {K}, B = {B}, kind {kind}, {dt.__name__}: mismatches so far = {bad}", flush=True)
print("FUZZ RESULT:", "clean" if bad == 0 else f"{bad} mismatches")
