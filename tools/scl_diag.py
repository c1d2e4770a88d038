"""Diagnostic: run the stamped SCL build (libechoseal_hip_diag.so) and a batch-size sweep."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1] == "diag":
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), "libechoseal_hip_diag.so")
from echoseal_amd.engine import RxEngine
eng = RxEngine(0)
rng = np.random.default_rng(0)
for B in (64, 1024, 4096, 16384, 65536):
    llr = torch.from_numpy(np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)).to(eng.device)
    eng.scl(llr, list_size=8); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.scl(llr, list_size=8); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} L=8 scl {dt*1e3:.2f} ms -> {B/dt:.0f} frames/s", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "diag" and B >= 1024: break
