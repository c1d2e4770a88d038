"""Host time to ENQUEUE one pipelined step (ctypes launches, torch allocations, events) against its GPU time."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
from echoseal_amd.engine import RxEngine, DecodePipeline
KEY = b"\xAA" * 32; B = 1024
eng = RxEngine(0, list_size_max=8); dev = eng.device
tx = WatermarkEmbedder(KEY); ctrs = list(range(B))
f = torch.from_numpy(tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))).to(dev)
b = torch.from_numpy(np.array([band_index(KEY, c) for c in ctrs], np.uint8)).to(dev)
p = torch.from_numpy(tx.sec.pn_bytes_batch(ctrs, 152)).to(dev)
pipe = DecodePipeline(eng, list_size=8)
for _ in range(5): pipe.submit(f, b, p)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): pipe.submit(f, b, p)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms/step of host time (thread blocked or not), total {1e3 * (t2 - t0) / n:.3f} ms/step")
# pure host cost: enqueue with the GPU far behind is not measurable directly (the depth throttle is device-side), so time tiny batches
f2, b2, p2 = f[:8].contiguous(), b[:8].contiguous(), p[:8].contiguous()
for _ in range(5): pipe.submit(f2, b2, p2)
torch.cuda.synchronize()
c0 = time.process_time(); t0 = time.perf_counter()
for _ in range(n): pipe.submit(f2, b2, p2)
c1 = time.process_time(); t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"8-frame batches: host CPU {1e3 * (c1 - c0) / n:.3f} ms/step, wall {1e3 * (t1 - t0) / n:.3f} ms/step")
