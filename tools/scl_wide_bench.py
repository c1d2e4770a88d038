"""Throughput and single-frame latency of the wide list decoder (L = 64, 128, 256).  argv[1] = library variant suffix (optional)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=256); rng = np.random.default_rng(0)
llr = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
for L in (64, 128, 256):
    B = 4096
    eng.scl(llr[:64], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.scl(llr[:B], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); eng.scl(llr[:1], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); d1 = time.perf_counter() - t1
    print(f"L={L:3d} B={B}: {dt * 1e3:8.1f} ms -> {B / dt:8.0f} frames/s ({B * L / dt / 1e6:.2f} M paths/s); one frame alone {d1 * 1e3:.2f} ms", flush=True)
