"""Launch es_xcorr32_kernel on a C3-sized batch (for rocprofv3 --pmc / --stats)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0); rng = np.random.default_rng(0)
for B in (65536, 1024):
    x = torch.from_numpy(rng.normal(0, 0.3, (B, 1215)).astype(np.float32)).to(eng.device)
    band = torch.from_numpy(rng.integers(0, 4, B).astype(np.uint8)).to(eng.device)
    y, y32 = eng.bpf2(x, band)
    for _ in range(4):
        c = eng.xcorr32(y32, band)
    torch.cuda.synchronize()
