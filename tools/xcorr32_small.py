"""Mean launch time of the float32 correlation screen at small batch sizes (HIP events, back-to-back launches)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), sys.argv[1])
from echoseal_amd.engine import RxEngine
eng = RxEngine(0); rng = np.random.default_rng(0)
for B in ((65536, 262144) if len(sys.argv) > 1 else (256, 1024, 2048, 4096, 16384, 65536)):
    x = torch.from_numpy(rng.normal(0, 0.3, (B, 1215)).astype(np.float32)).to(eng.device)
    band = torch.from_numpy(rng.integers(0, 4, B).astype(np.uint8)).to(eng.device)
    y, y32 = eng.bpf2(x, band)
    for _ in range(10):
        c = eng.xcorr32(y32, band)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        c = eng.xcorr32(y32, band)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"B={B:6d}  {us:8.2f} us/launch  {B * 9472 / us / 1e3:8.1f} GB/s", flush=True)
