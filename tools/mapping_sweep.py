"""Which list-decoder mapping is fastest at which batch size (L = 8 unless argv[1])?  One line per batch size: one frame per wave,
four / two lanes per path (several frames per wave), one lane per path, and what es_scl_batch picks by itself."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = RxEngine(0, list_size_max=max(8, L)); eng.set_option("scl_lane_slab", 1)
rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
def t(llr):
    eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
for B in (512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 65536):
    llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
    out = []
    for multi, lanes in ((0, 4), (1, 4), (1, 2), (1, 1), (-1, 0)):
        if L > 16 and (multi, lanes) == (1, 4):
            out.append(float("nan")); continue
        eng.set_option("scl_multi", multi); eng.set_option("scl_lanes", lanes)
        out.append(t(llr) * 1e3)
    eng.set_option("scl_multi", -1); eng.set_option("scl_lanes", 0)
    best = int(np.nanargmin(out[:4]))
    print(f"L={L} B={B:6d}: frame/wave {out[0]:7.2f}  4 lanes {out[1]:7.2f}  2 lanes {out[2]:7.2f}  1 lane {out[3]:7.2f}  auto {out[4]:7.2f} ms   best: {('frame/wave', '4 lanes', '2 lanes', '1 lane')[best]}"
          f"{'' if out[4] <= 1.03 * out[best] else '   <-- auto is %.0f %% slower' % (100 * (out[4] / out[best] - 1))}", flush=True)
