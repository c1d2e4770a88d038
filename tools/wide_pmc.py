"""Warm launches of the lane-per-path list decoder for rocprofv3 --pmc: python3 tools/wide_pmc.py [L] [B] [NAME]  (L <= 32: scl_lanes = 1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 3 and sys.argv[3]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[3]}.so")
from echoseal_amd.engine import RxEngine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
eng = RxEngine(0, list_size_max=256)
eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", 1)
rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
for _ in range(2):
    eng.scl(llr, list_size=L, skip_if_hard_ok=False)
torch.cuda.synchronize()
