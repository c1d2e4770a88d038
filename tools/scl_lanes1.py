"""One lane per path (es_scl_wide.hip, scl_lanes = 1) against the multi-frame kernel: identical results, time by batch size."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=64); rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (16384, 1024)), -12, 12).astype(np.float32)).to(eng.device)
base[::9] = torch.clamp(base[::9] * 6, -12, 12)
Ls = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 1, 2, 4, 16, 32, 24, 5]
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1000, 16384, 65536]
for L in Ls:
    for B in Bs:
        llr = base.repeat(-(-B // 16384), 1)[:B].contiguous()
        out = {}
        for lanes in (0, 1):
            eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", lanes)
            r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
            t0 = time.perf_counter(); r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out[lanes] = (dt, r)
        same = all(torch.equal(getattr(out[0][1], k), getattr(out[1][1], k)) for k in ("hard_info", "hard_ok", "cand_info", "cand_metric", "cand_ok", "ncand"))
        print(f"L={L:2d} B={B:6d}: multi {out[0][0] * 1e3:8.2f} ms ({B / out[0][0] / 1e6:.3f} M/s)  lane-per-path {out[1][0] * 1e3:8.2f} ms ({B / out[1][0] / 1e6:.3f} M/s)  identical={same}", flush=True)
