"""es_xcorr32_batch on the BASELINE config-3 shape (65 536 windows of 2 048 samples): numerators on the matrix pipe
(es_xcorr32_mfma_kernel, option xcorr_mfma = 1) against the packed-vector
kernel (es_xcorr32_kernel<17,2048>, option 0).
Prints the launch time of each (HIP events on the engine's stream, median of 20 warm launches), the largest difference between the
two screens, each screen's largest distance from the float64 correlation (es_xcorr_batch; the picker's bound is DELTA = 3e-5), and
whether es_pick_exact_batch settles every record identically from either screen.   python3 tools/xcorr32_mfma_ab.py [B]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import echoseal_amd._native as nat
if os.environ.get("ES_LIB_VARIANT"):                  # tools/build_variant.sh NAME es_sync32.hip "-D..."
    nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{os.environ['ES_LIB_VARIANT']}.so")
from echoseal_amd.engine import RxEngine
from echoseal_amd import workloads as WL
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = RxEngine(0, list_size_max=8)
fr, _ = eng.synthetic_frames(KEY, 0, B)
band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=B)[1]
win, off = WL.c3_windows_device(fr)
y, y32 = eng.bpf2(win, band)
c64 = eng.xcorr(y[:8192], band[:8192])
out = {}
for mode in (0, 1, 0, 1):
    eng.set_option("xcorr_mfma", mode)
    for _ in range(3):
        c = eng.xcorr32(y32, band)
    ts = []
    for _ in range(5):                                # 20 launches back to back between two events: launch overhead hidden
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            c = eng.xcorr32(y32, band)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ms = float(np.median(ts))
    ts = []
    for _ in range(8):                                # as bench.py's c3_unfused leg times it: right behind the band-pass that wrote y32
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        yy, yy32 = eng.bpf2(win, band)
        e0.record(); c = eng.xcorr32(yy32, band); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms_situ = float(np.mean(ts[2:]))
    if os.environ.get("ES_AB_TIMING_ONLY"):
        print(f"xcorr_mfma={mode}: behind the band-pass {ms_situ:.4f} ms; back to back {ms:.4f} ms", flush=True); continue
    ee = (c[:8192].double() - c64).abs(); err = float(ee[torch.isfinite(ee)].max())
    p = eng.pick_exact(c, y, band)
    out[mode] = (c, p)
    print(f"xcorr_mfma={mode}: behind the band-pass {ms_situ:.4f} ms = {16136 * B / ms_situ / 8e9:.3f} of 8 TB/s; back to back {ms:.4f} ms  ({16136 * B / ms / 1e6:.0f} GB/s algorithmic, {16136 * B / ms / 8e9:.3f} of 8 TB/s)  "
          f"max|screen - float64| = {err:.2e}  nan = {int(torch.isnan(c).sum())}", flush=True)
if os.environ.get("ES_AB_TIMING_ONLY"): sys.exit(0)
for m in (1,):
    dd = (out[0][0] - out[m][0]).abs()
    d = float(dd[torch.isfinite(dd)].max())
    same = all(torch.equal(u, v) for u, v in zip(out[0][1], out[m][1]))
    print(f"mode {m}: max|matrix - packed| = {d:.2e}; NaN rows {int(torch.isnan(out[m][0]).any(dim=1).sum())}; flagged records "
          f"{int((out[m][1][3] != 0).sum())} (packed: {int((out[0][1][3] != 0).sum())}); pick_exact results identical: {same}")
