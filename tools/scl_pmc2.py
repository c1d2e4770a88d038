"""One warm launch pair of each several-frames-per-wave mapping of the list decoder (L = 8) at B = 65 536 for rocprofv3 --pmc (optionally another build:
python3 tools/scl_pmc2.py NAME)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=8)
rng = np.random.default_rng(0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
eng.set_option("scl_multi", 1)
for lanes in (4, 2, 1):                  # 16 paths x 4 lanes per wave, 32 paths x 2 lanes, 64 paths x 1 lane (es_scl_wide.hip: the grouped pipeline, large batches)
    eng.set_option("scl_lanes", lanes)
    for _ in range(2):
        eng.scl(llr, list_size=8, skip_if_hard_ok=False)
torch.cuda.synchronize()
