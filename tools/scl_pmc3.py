"""Two warm launches of the one-lane-per-path list decoder (es_scl_wide_kernel<64,8>, L = 8) at B = 65 536 for rocprofv3 --pmc
(round 3: FETCH_SIZE / WRITE_SIZE / per-type VALU counters of the kernel that dominates the time).
    python3 tools/scl_pmc3.py [BUILD_NAME] [B] [L] [MAPPING]      MAPPING: lane (default) | frame (one frame per wave, es_scl_kernel<L>)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = RxEngine(0, list_size_max=max(8, L))
rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
if len(sys.argv) > 4 and sys.argv[4] == "frame":
    eng.set_option("scl_multi", 0)
else:
    eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", 1)
for _ in range(3):
    eng.scl(llr, list_size=L, skip_if_hard_ok=False)
torch.cuda.synchronize()
