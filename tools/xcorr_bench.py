"""Launch es_xcorr_kernel a few times on a C3-sized (65 536 x 1215) and a C2-sized (1 024) batch.
Used under rocprofv3 (--kernel-trace --stats, or separate --pmc passes) to price the kernel."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0)
rng = np.random.default_rng(0)
for B in (65536, 1024):
    x = torch.from_numpy(rng.normal(0, 0.3, (B, 1215)).astype(np.float32)).to(eng.device)
    band = torch.from_numpy(rng.integers(0, 4, B).astype(np.uint8)).to(eng.device)
    y = eng.bpf(x, band)
    eng.xcorr(y, band); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        c = eng.xcorr(y, band)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"B={B}: xcorr {dt*1e3:.3f} ms/launch  {B/dt/1e6:.1f} M records/s  algorithmic {9472*B/dt/1e9:.0f} GB/s  actual(f64 io) {18944*B/dt/1e9:.0f} GB/s", flush=True)
    t0 = time.perf_counter()
    for _ in range(5):
        thr, pk, npk = eng.pick(c)
    torch.cuda.synchronize()
    print(f"      pick {(time.perf_counter()-t0)/5*1e3:.3f} ms/launch", flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        y = eng.bpf(x, band)
    torch.cuda.synchronize()
    print(f"      bpf {(time.perf_counter()-t0)/3*1e3:.3f} ms/launch", flush=True)
