"""Throughput of es_schedule_batch (device-side key/PN/hop schedule) against the host code."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
from echoseal_amd.crypto import SecureChannel
from echoseal_amd.dist import build_schedule
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8); sec = SecureChannel(KEY)
n = 1 << 20
eng.schedule(sec._prng.sub_key, KEY, ctr0=0, n=n); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): eng.schedule(sec._prng.sub_key, KEY, ctr0=0, n=n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"device: {n} counters in {ms:.3f} ms -> {n / ms / 1e3:.1f} M counters/s ({n * 153 / ms / 1e6:.1f} GB/s of schedule)")
t0 = time.perf_counter(); build_schedule(KEY, range(16384)); dt = time.perf_counter() - t0
print(f"host (numpy AES + hmac): 16384 counters in {dt * 1e3:.1f} ms -> {16384 / dt / 1e3:.1f} k counters/s")
# frame generator (f-3): 65 536 frames on the device vs the host embedder
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
tx = WatermarkEmbedder(KEY); nf = 65536
pl = torch.randint(0, 256, (nf, 55), dtype=torch.uint8)
ct = torch.arange(nf)
eng.make_frames(tx.sec, KEY, ct, pl); torch.cuda.synchronize()
t0 = time.perf_counter(); fr = eng.make_frames(tx.sec, KEY, ct, pl); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"device frame generator: {nf} frames in {dt * 1e3:.2f} ms -> {nf / dt / 1e6:.2f} M frames/s")
c = list(range(256)); p = synthetic_payloads(tx.sec, c)
t0 = time.perf_counter(); tx.make_frames(c, p); dt = time.perf_counter() - t0
print(f"host embedder: 256 frames in {dt * 1e3:.1f} ms -> {256 / dt:.0f} frames/s")
