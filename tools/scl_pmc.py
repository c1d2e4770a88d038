"""Launch the two list-decoder kernels once warm (for rocprofv3 --pmc): es_scl_kernel<8> on 1 024 frames (one wave per
SIMD) and es_scl_multi_kernel<8> on 16 384 frames (two waves per SIMD)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=8)
rng = np.random.default_rng(0)
for B in (1024, 16384):
    llr = torch.from_numpy(np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)).to(eng.device)
    for _ in range(3):
        eng.scl(llr, list_size=8, skip_if_hard_ok=False)
    torch.cuda.synchronize()
