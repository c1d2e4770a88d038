"""Large batches through the lane pipeline: does the front end of chunk k+1 overlap the list decoder of chunk k?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine, DecodePipeline
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8); dev = eng.device
n, chunk = 1 << 19, int(sys.argv[2]) if len(sys.argv) > 2 else 131072
frames = torch.cat([eng.synthetic_frames(KEY, k, 65536)[0] for k in range(0, n, 65536)])
pn, band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=n)
for lanes in (2, 3, 4):
    pipe = DecodePipeline(eng, list_size=8, lanes=lanes)
    for e in pipe.scl_engs: e.set_option("scl_multi", 1)
    def run():
        outs = []
        for c0 in range(0, n, chunk):
            sy, llr, scl, done = pipe.submit(frames[c0:c0 + chunk], band[c0:c0 + chunk], pn[c0:c0 + chunk])
            outs.append(scl)
        torch.cuda.synchronize()
        return outs
    run()
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    print(f"[{sys.argv[1] if len(sys.argv) > 1 else 'product'}] chunk {chunk} lanes {lanes}: {n / dt / 1e3:.1f} k frames/s", flush=True)
    del pipe
