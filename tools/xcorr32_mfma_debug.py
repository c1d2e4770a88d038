"""Where does the matrix-pipe screen (xcorr_mfma = 1; or an experimental build of another mode) differ from the packed-vector one?  Histograms of the differing lags by segment, tile,
position in the tile; python3 tools/xcorr32_mfma_debug.py [mode]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if os.environ.get("ES_LIB_VARIANT"):
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{os.environ['ES_LIB_VARIANT']}.so")
from echoseal_amd.engine import RxEngine
from echoseal_amd import workloads as WL
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 4096
eng = RxEngine(0, list_size_max=0)
fr, _ = eng.synthetic_frames(KEY, 0, B)
band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=B)[1]
win, off = WL.c3_windows_device(fr)
y, y32 = eng.bpf2(win, band)
eng.set_option("xcorr_mfma", 0); c0 = eng.xcorr32(y32, band)
eng.set_option("xcorr_mfma", mode); c2 = eng.xcorr32(y32, band); c2b = eng.xcorr32(y32, band)
torch.cuda.synchronize()
print("repeat launches identical:", bool(torch.equal(c2, c2b)))
d = (c2 - c0).abs()
bad = (d > 1e-4) | torch.isnan(d)
print("records with a differing lag:", int(bad.any(dim=1).sum()), "of", B, "; differing lags:", int(bad.sum()), "of", bad.numel(), "; max", float(d[~torch.isnan(d)].max()))
lag = torch.arange(bad.shape[1], device=bad.device)[None, :].expand_as(bad)[bad]
rec = torch.arange(B, device=bad.device)[:, None].expand_as(bad)[bad]
for name, v, n in (("segment", lag // 1024, 2), ("tile", (lag % 1024) // 256, 4), ("lag % 16", lag % 16, 16), ("(lag % 256) // 16", (lag % 256) // 16, 16),
                   ("band", band[rec].long(), 4), ("record % 8", rec % 8, 8)):
    print(name, torch.bincount(v, minlength=n).tolist())
r = int(rec[0]) if len(rec) else 0
print("record", r, "band", int(band[r]), "first lags  packed:", c0[r, :8].tolist(), " matrix:", c2[r, :8].tolist())
print("ratio matrix / packed on that record (median):", float((c2[r] / c0[r]).median()))
