"""The list decoder's mappings on the config-5 surrogate's LLRs (a share of the frames pass the hard-decision shortcut and skip the list):
per list size, milliseconds of es_scl_batch(skip_if_hard_ok=1) under each mapping, and what it picks by itself."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
import echoseal_amd.workloads as WL
eng = RxEngine(0, list_size_max=64); dev = eng.device
U, Bn = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frames, band, pn, payloads = WL.c2_frames(range(U))
lossy = WL.lossy_channel(frames)
reps = -(-Bn // U)
f = torch.from_numpy(lossy).to(dev).repeat(reps, 1)[:Bn].contiguous()
b = torch.from_numpy(band).to(dev).repeat(reps)[:Bn].contiguous()
p = torch.from_numpy(pn).to(dev).repeat(reps, 1)[:Bn].contiguous()
sy, llr, scl = eng.decode_batch(f, b, p, list_size=8)
torch.cuda.synchronize()
print("frames", Bn, " hard-decision ok:", int(scl.hard_ok.sum()), " |llr| max", float(llr.abs().max()), "mean", float(llr.abs().mean()))
def t(L, skip):
    eng.scl(llr, list_size=L, skip_if_hard_ok=skip); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.scl(llr, list_size=L, skip_if_hard_ok=skip); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for L in (1, 4, 8, 16):
    for skip in (True, False):
        out = []
        for multi, lanes in ((0, 4), (1, 4), (1, 2), (1, 1), (-1, 0)):
            if (L > 16 and (multi, lanes) == (1, 4)):
                out.append(float("nan")); continue
            eng.set_option("scl_multi", multi); eng.set_option("scl_lanes", lanes)
            out.append(t(L, skip))
        eng.set_option("scl_multi", -1); eng.set_option("scl_lanes", 0)
        print(f"L={L:2d} skip={int(skip)}: frame/wave {out[0]:7.2f}  4 lanes {out[1]:7.2f}  2 lanes {out[2]:7.2f}  1 lane {out[3]:7.2f}  auto {out[4]:7.2f} ms", flush=True)
