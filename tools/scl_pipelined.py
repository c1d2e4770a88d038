"""SCL-8 alone, pipelined: N engines on N streams, each launching 65 536-frame batches back to back (no front end) -- the steady-state rate of
the list decoder when the tails of one launch are filled by the next.  python3 tools/scl_pipelined.py [LANES] [BUILD]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import echoseal_amd._native as nat
if len(sys.argv) > 2 and sys.argv[2]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[2]}.so")
from echoseal_amd.engine import RxEngine
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for sigma in (3.0, 0.75):
    rnd = torch.clamp(sigma * torch.randn((65536, 1024), device=dev, generator=g), -12, 12)
    engs = [RxEngine(0, list_size_max=8) for _ in range(lanes)]
    for e in engs: e.set_option("scl_lane_slab", 1)
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
    def run(rounds):
        for r in range(rounds):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.scl(rnd, list_size=8, skip_if_hard_ok=True)
        torch.cuda.synchronize()
    run(1)
    t0 = time.perf_counter(); run(5); dt = time.perf_counter() - t0
    print(f"sigma {sigma}: {lanes} lanes x 5 launches of 65 536: {lanes * 5 * 65536 / dt / 1e6:.3f} M frames/s ({dt / (lanes * 5) * 1e3:.2f} ms per launch)", flush=True)
    for e in engs: e.close()
