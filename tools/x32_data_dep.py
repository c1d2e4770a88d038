"""Is the correlation screen's launch time data dependent?  Same shape (65 536 x 2 048), different contents."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=8); dev = eng.device
B, W = 65536, 2048
g = torch.Generator(device=dev); g.manual_seed(4)
band = torch.randint(0, 4, (B,), device=dev, generator=g).to(torch.uint8)
def run(name, x, bb):
    ms = []
    for it in range(5):
        y64, y32 = eng.bpf2(x, bb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = eng.xcorr32(y32, bb); e1.record(); torch.cuda.synchronize()
        if it: ms.append(e0.elapsed_time(e1))
        del y64, y32, c
    print(f"{name:34s} {np.mean(ms):.3f} ms", flush=True)
noise = torch.randn((B, W), device=dev, generator=g)
def run2(name, x, bb, follow):
    evs = []
    for it in range(6):
        y64, y32 = eng.bpf2(x, bb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = eng.xcorr32(y32, bb); e1.record()
        if follow: r = eng.pick_exact(c, y64, bb)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    print(f"{name:34s} {np.mean([a.elapsed_time(b) for a, b in evs[2:]]):.3f} ms", flush=True)
run2("no sync, nothing after", noise * 0.5, band, False)
run2("no sync, pick_exact after", noise * 0.5, band, True)
run("distinct noise sigma 0.05", noise * 0.05, band)
run("distinct noise sigma 0.5", noise * 0.5, band)
run("512 distinct rows tiled", (noise[:512] * 0.5).repeat(128, 1), band)
run("512 rows tiled, band tiled", (noise[:512] * 0.5).repeat(128, 1), band[:512].repeat(128))
run("distinct noise, band all 0", noise * 0.5, torch.zeros_like(band))
run("zeros", torch.zeros_like(noise), band)
