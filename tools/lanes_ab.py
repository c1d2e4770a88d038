"""A/B of a library variant on the lane-per-path kernel: python tools/lanes_ab.py NAME [L] [B]: time + equality with the multi-frame kernel."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 73728
eng = RxEngine(0, list_size_max=64); rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (8192, 1024)), -12, 12).astype(np.float32)).to(eng.device)
base[::9] = torch.clamp(base[::9] * 6, -12, 12)
llr = base.repeat(-(-B // 8192), 1)[:B].contiguous()
eng.set_option("scl_multi", 1)
ref = eng.scl(llr, list_size=L, skip_if_hard_ok=False)
eng.set_option("scl_lanes", 1)
ts = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    same = all(torch.equal(getattr(ref, k), getattr(r, k)) for k in ("hard_info", "hard_ok", "cand_info", "cand_metric", "cand_ok", "ncand"))
    print(f"{sys.argv[1] or 'base':8s} L={L} B={B} run {rep}: {ts[-1] * 1e3:8.2f} ms  {B / ts[-1] / 1e6:.3f} M frames/s  identical={same}", flush=True)
