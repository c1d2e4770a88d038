"""Diagnostic builds of the float16-pair experiment (ES_LIB_VARIANT=diag1: numerators alone; diag2: normalisation factors alone) against torch:
which of the two carries the wrong rows?  python3 tools/xcorr32_mfma16_diag.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
v = os.environ["ES_LIB_VARIANT"]
nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{v}.so")
from echoseal_amd.engine import RxEngine
from echoseal_amd import workloads as WL
from echoseal_amd.embedder import WatermarkEmbedder
from echoseal_amd.tables import pack_tables
KEY = b"\xAA" * 32
B = 4096
eng = RxEngine(0, list_size_max=0)
fr, _ = eng.synthetic_frames(KEY, 0, B)
band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=B)[1]
win, off = WL.c3_windows_device(fr)
y, y32 = eng.bpf2(win, band)
tpl = torch.from_numpy(pack_tables()[1].astype(np.float32)).to(y32.device)           # [4][63]
yd = y32.double()
if v == "diag1":
    ref = torch.empty((B, 1986), dtype=torch.float64, device=y32.device)
    for b in range(4):
        m = band == b
        ref[m] = torch.nn.functional.conv1d(yd[m][:, None, :], tpl[b].double()[None, None, :])[:, 0, :]
    scale = ref.abs().max()
else:
    en = torch.nn.functional.conv1d((yd * yd)[:, None, :], torch.ones(1, 1, 63, dtype=torch.float64, device=y32.device))[:, 0, :]
    ref = en.rsqrt(); scale = 1.0
eng.set_option("xcorr_mfma", 2)
for rep in range(3):
    c = eng.xcorr32(y32, band).double()
    d = ((c - ref) / (ref.abs() if v == "diag2" else scale)).abs()
    bad = d > 1e-3
    lag = torch.arange(1986, device=bad.device)[None, :].expand_as(bad)[bad]
    print(v, "rep", rep, "bad lags", int(bad.sum()), "records", int(bad.any(dim=1).sum()), "typical rel err", float(d[~bad].max()),
          "tile", torch.bincount((lag % 1024) // 256, minlength=4).tolist(), "lag%16", torch.bincount(lag % 16, minlength=16).tolist(), flush=True)
    if rep == 0 and int(bad.sum()):
        rec = torch.arange(B, device=bad.device)[:, None].expand_as(bad)[bad]
        r0 = int(rec[0]); lags = lag[rec == r0]
        print("record", r0, "bad lags", lags.tolist()[:40])
        for L in lags.tolist()[:6]:
            val = float(c[r0, L]); near = (ref[r0] - val).abs() / ref[r0].abs()
            print("  lag", L, "got", val, "want", float(ref[r0, L]), "ratio", val / float(ref[r0, L]), "closest reference lag in the record:", int(near.argmin()), float(near.min()),
                  "neighbours got/want", [(round(float(c[r0, L + d]), 4), round(float(ref[r0, L + d]), 4)) for d in (-2, -1, 1, 2)])
