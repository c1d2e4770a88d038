"""Warm launches of es_xcorr32_batch on the BASELINE config-3 shape (65 536 windows of 2 048 samples) with the numerators on the
packed vector FMAs (option xcorr_mfma = 0) and on the matrix pipe (1), each right behind the band-pass as in bench.py's c3_unfused
leg: for rocprofv3 --pmc / --kernel-trace (tools/collect_pmc_xmfma.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine
from echoseal_amd import workloads as WL
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=0)
fr, _ = eng.synthetic_frames(KEY, 0, 65536)
band = eng.schedule(WatermarkEmbedder(KEY).sec._prng.sub_key, KEY, ctr0=0, n=65536)[1]
win, off = WL.c3_windows_device(fr)
for mode in (0, 1):
    eng.set_option("xcorr_mfma", mode)
    for _ in range(4):
        y, y32 = eng.bpf2(win, band)
        c = eng.xcorr32(y32, band)
torch.cuda.synchronize()
