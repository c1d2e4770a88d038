"""Throughput of the f-2 kernels: AEAD validator on 2^20 blobs, selection over [65536, 8] and [4096, 256] lists."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine, SclResult
eng = RxEngine(0, list_size_max=8); dev = eng.device
g = torch.Generator(device="cpu"); g.manual_seed(1)
key = bytes(range(32))
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 1 << 20
blobs = torch.randint(0, 256, (N, 55), dtype=torch.uint8, generator=g).to(dev)
ctrs = torch.randint(0, 2 ** 31, (N,), dtype=torch.int64, generator=g)
cd = eng._ctr_dev(ctrs)
ms = timed(lambda: eng.aead_check(key, blobs, cd.to(torch.int64) & 0xFFFFFFFF))
print(f"aead_check: {N} blobs in {ms:.3f} ms -> {N / ms / 1e3:.1f} M blobs/s, {N * 55 / ms / 1e6:.1f} GB/s of blob bytes", flush=True)
for B, L in ((65536, 8), (4096, 256)):
    res = SclResult(torch.randint(0, 256, (B, 55), dtype=torch.uint8, generator=g).to(dev), torch.zeros(B, dtype=torch.uint8, device=dev),
                    torch.randint(0, 256, (B, L, 55), dtype=torch.uint8, generator=g).to(dev),
                    torch.sort(torch.rand((B, L), dtype=torch.float64, generator=g), dim=1).values.to(dev),
                    (torch.rand((B, L), generator=g) < 1 / 256).to(torch.uint8).to(dev), torch.full((B,), L, dtype=torch.int32, device=dev))
    c = ctrs[:B]
    ms = timed(lambda: eng.select(res, key32=key, ctrs=c))
    print(f"select: B={B} L={L} in {ms:.3f} ms -> {B / ms / 1e3:.2f} M frames/s", flush=True)
