"""The stages of one headline (C3) step run alone, one after the other on one stream: 65 536 windows of 2 048 as bench.py builds them
(clean frames resampled +-5 %, offset, in noise at -15 dB), front end (band-pass + fused sync + LLR) and SCL-8 on ITS LLRs."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import echoseal_amd._native as nat
if len(sys.argv) > 1 and sys.argv[1]:
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{sys.argv[1]}.so")
from echoseal_amd.engine import RxEngine
from echoseal_amd.embedder import WatermarkEmbedder
import echoseal_amd.workloads as WL
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8); eng.set_option("scl_lane_slab", 1); dev = eng.device
Bw = 65536
clean = torch.cat([eng.synthetic_frames(KEY, c0, 16384)[0] for c0 in range(0, Bw, 16384)])
win, off = WL.c3_windows_device(clean, seed=34)
pn, band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=Bw)
def best(fn, n=3):
    fn(); torch.cuda.synchronize(); b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3, r
def front():
    y, y32 = eng.bpf2(win, band)
    thr, peaks, npeaks, flags = eng.sync_fused(y, y32, band)
    return eng.llr(y, band, pn, start=peaks[:, 0].clamp(min=0).contiguous(), variant=0)
t_front, llr = best(front)
y_, y32_ = eng.bpf2(win, band); thr_, peaks_, np_, fl_ = eng.sync_fused(y_, y32_, band); st_ = peaks_[:, 0].clamp(min=0).contiguous()
t_bpf, _ = best(lambda: eng.bpf2(win, band)); t_sync, _ = best(lambda: eng.sync_fused(y_, y32_, band)); t_llr, _ = best(lambda: eng.llr(y_, band, pn, start=st_, variant=0))
print(f"   band-pass {t_bpf:.3f} ms, fused sync {t_sync:.3f} ms, demodulator {t_llr:.3f} ms")
t_scl, scl = best(lambda: eng.scl(llr, list_size=8, skip_if_hard_ok=True))
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = torch.clamp(3.0 * torch.randn(llr.shape, device=dev, generator=g), -12, 12)
t_rnd, _ = best(lambda: eng.scl(rnd, list_size=8, skip_if_hard_ok=True))
print(f"[{sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else 'product'}] headline windows: front end {t_front:.2f} ms, SCL-8 on their LLRs {t_scl:.2f} ms ({Bw / t_scl / 1e3:.2f} M frames/s; hard-decision shortcut taken by {int((scl.ncand == 0).sum())} of {Bw}; "
      f"mean |LLR| {float(llr.abs().mean()):.2f}); SCL-8 on N(0, 3) rows {t_rnd:.2f} ms ({Bw / t_rnd / 1e3:.2f} M frames/s)")
