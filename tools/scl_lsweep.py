"""SCL throughput by list size at a large batch (random LLRs): every kernel family, frames/s and paths/s."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=256); rng = np.random.default_rng(0)
B = 16384
llr = torch.from_numpy(np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)).to(eng.device)
for multi in (0, 1):
    eng.set_option("scl_multi", multi)
    for L in ((1, 2, 4, 8, 16, 32, 64, 128, 256) if multi == 0 else (1, 2, 4, 8, 16)):
        n = B if L <= 32 else B // 4
        eng.scl(llr[:n], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.scl(llr[:n], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"multi={multi} L={L:3d} B={n}: {dt * 1e3:8.2f} ms -> {n / dt:10.0f} frames/s  {n * L / dt / 1e6:6.2f} M paths/s", flush=True)
