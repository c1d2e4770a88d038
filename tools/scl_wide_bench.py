"""Throughput of the wide list decoder (L = 64, 128, 256)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=256); rng = np.random.default_rng(0)
for L, B in ((64, 2048), (128, 1024), (256, 1024)):
    llr = torch.from_numpy(np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)).to(eng.device)
    eng.scl(llr[:64], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); eng.scl(llr[:1], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); d1 = time.perf_counter() - t1
    print(f"L={L:3d} B={B}: {dt * 1e3:8.1f} ms -> {B / dt:8.0f} frames/s; one frame alone {d1 * 1e3:.1f} ms", flush=True)
