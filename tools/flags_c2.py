"""How many C2 records does the float32 screen hand to the float64 redo kernels, and why (flag reason codes)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.dist import build_schedule, split_schedule
from echoseal_amd.engine import RxEngine
KEY = bytes([0xAA]) * 32
B = 1024
tx = WatermarkEmbedder(KEY); ctrs = list(range(B))
frames = torch.from_numpy(tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))).cuda()
sched = torch.from_numpy(build_schedule(KEY, range(B))).cuda() if not torch.is_tensor(build_schedule(KEY, range(B))) else build_schedule(KEY, range(B)).cuda()
pn, band = split_schedule(sched, 0, B)
eng = RxEngine(0)
y, y32 = eng.bpf2(frames, band)
c32 = eng.xcorr32(y32, band)
thr, peaks, npeaks, flags = eng.pick_exact(c32, y, band)
f = flags.cpu().numpy()
print("flag histogram:", {int(k): int(v) for k, v in zip(*np.unique(f, return_counts=True))})
c64 = eng.xcorr(y, band).cpu().numpy(); c = c32.cpu().numpy()
print("max |corr32 - corr64| =", np.nanmax(np.abs(c - c64)))
idx = np.flatnonzero(f)
for i in idx[:5]:
    print("record", i, "flag", f[i], "nan in corr32:", int(np.isnan(c[i]).sum()), "min/max y32", float(y32[i].min()), float(y32[i].max()))
