"""Do several list-decoder launches on different streams (each with its own context) really overlap?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
rng = np.random.default_rng(0); B = 1024
engs = [RxEngine(0, list_size_max=8) for _ in range(4)]
dev = engs[0].device
llr = torch.from_numpy(np.clip(rng.normal(0, 3, (B, 1024)), -12, 12).astype(np.float32)).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(4)]
for multi in (0, 1):
    for e in engs: e.set_option("scl_multi", multi)
    for n in (1, 2, 3, 4):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                with torch.cuda.stream(streams[k]):
                    engs[k].scl(llr, list_size=8, skip_if_hard_ok=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"multi={multi}: {n} concurrent launches of {B} frames: {dt * 1e3:.2f} ms -> {n * B / dt / 1e3:.0f} k frames/s", flush=True)
