"""WatermarkDetector.verify() on a 5 s clip (the f-1 / f-2 flow: four band scans, header decode, counter search,
4 x list-256 decodes per candidate, AEAD validation -- all on the GPU): wall time per call."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder
from echoseal_amd.detector import WatermarkDetector
KEY = b"\xAA" * 32
rng = np.random.default_rng(1)
host = (0.05 * rng.standard_normal(5 * 48000)).astype(np.float32)
tx = WatermarkEmbedder(KEY)
audio = tx.process(host)
for L in (8, 256):
    det = WatermarkDetector(KEY, list_size=L)
    det.verify(audio[:48000], 48000); torch.cuda.synchronize()
    t0 = time.perf_counter(); ok = det.verify(audio, 48000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"list_size={L:3d}: verify(5 s clip) -> {ok} in {dt:.2f} s", flush=True)
