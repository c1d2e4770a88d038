"""Warm launches of the correlation kernels on the BASELINE config-3 shape (65 536 windows of 2 048 samples) for
rocprofv3 --pmc / --stats: es_xcorr32_kernel<17,2048> (stand-alone screen) and the fused sync kernel."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
from echoseal_amd import workloads as WL
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8)
fr, _ = eng.synthetic_frames(KEY, 0, 65536)
band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=65536)[1]
win, off = WL.c3_windows_device(fr)
y, y32 = eng.bpf2(win, band)
for _ in range(3):
    c = eng.xcorr32(y32, band)
for _ in range(3):
    r = eng.sync_fused(y, y32, band)
torch.cuda.synchronize()
