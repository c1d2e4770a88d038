"""End-to-end (sync + LLR + SCL-8) throughput versus batch size on one GPU."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
KEY = b"\xAA" * 32
tx = WatermarkEmbedder(KEY)
U = 1024
ctrs = list(range(U))
frames = tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))
band = np.array([band_index(KEY, c) for c in ctrs], np.uint8)
pn = tx.sec.pn_bytes_batch(ctrs, 152)
eng = RxEngine(0)
d = eng.device
for B in (1024, 4096, 16384, 65536, 131072):
    r = B // U
    f = torch.from_numpy(frames).to(d).repeat(r, 1); b = torch.from_numpy(band).to(d).repeat(r); p = torch.from_numpy(pn).to(d).repeat(r, 1)
    eng.decode_batch(f, b, p, list_size=8); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        eng.decode_batch(f, b, p, list_size=8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt*1e3:.2f} ms/step -> {B/dt:.0f} frames/s", flush=True)
for L in (64, 256):
    eng2 = RxEngine(0, list_size_max=256)
    llr = torch.from_numpy(np.clip(np.random.default_rng(0).normal(0, 3, (512, 1024)), -12, 12).astype(np.float32)).to(d)
    eng2.scl(llr, list_size=L); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng2.scl(llr, list_size=L); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"SCL-{L}: 512 frames in {dt*1e3:.1f} ms -> {512/dt:.0f} frames/s", flush=True)
    eng2.close()
