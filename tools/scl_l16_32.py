"""L = 16 / 32: one frame per wave (es_scl_kernel) against the multi-frame kernel family, by batch size; results compared."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=32); rng = np.random.default_rng(0)
llr = torch.from_numpy(np.clip(rng.normal(0, 3, (16384, 1024)), -12, 12).astype(np.float32)).to(eng.device)
for L in (16, 32, 24):
    for B in (64, 256, 1024, 4096, 16384):
        out = {}
        for multi in (0, 1):
            eng.set_option("scl_multi", multi)
            r = eng.scl(llr[:B], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
            t0 = time.perf_counter(); r = eng.scl(llr[:B], list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out[multi] = (dt, r)
        same = all(torch.equal(getattr(out[0][1], k), getattr(out[1][1], k)) for k in ("cand_info", "cand_metric", "cand_ok", "ncand"))
        print(f"L={L:2d} B={B:5d}: one-frame {out[0][0] * 1e3:8.2f} ms  multi {out[1][0] * 1e3:8.2f} ms  identical={same}", flush=True)
