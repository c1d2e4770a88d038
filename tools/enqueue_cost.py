"""CPU cost of enqueueing one pipeline step (lanes mode) vs the GPU time per step."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from echoseal_amd import workloads as WL
from echoseal_amd.engine import RxEngine, DecodePipeline
eng = RxEngine(0, list_size_max=8); dev = eng.device
frames, band, pn, _ = WL.c2_frames(range(1024))
f, b, p = (torch.from_numpy(x).to(dev) for x in (frames, band, pn))
for lanes in (7,):
    pipe = DecodePipeline(eng, list_size=8, lanes=lanes)
    for e in pipe.scl_engs: e.set_option("scl_multi", 1)
    for _ in range(14): pipe.submit(f, b, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): pipe.submit(f, b, p)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"lanes {lanes}: enqueue {1e3 * (t1 - t0) / 200:.3f} ms/step, total {1e3 * (t2 - t0) / 200:.3f} ms/step")
