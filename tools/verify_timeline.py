"""Where one verify() of a 5 s clip spends its time: wall-clock marks around the stages of WatermarkDetector.verify_batch (host clock,
synchronising after each stage -- the marks serialise what normally overlaps, so the sum exceeds an unmarked call)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder
from echoseal_amd.detector import WatermarkDetector
import cProfile, pstats, io
KEY = b"\xAA" * 32
rng = np.random.default_rng(1)
audio = WatermarkEmbedder(KEY).process((0.05 * rng.standard_normal(5 * 48000)).astype(np.float32))
det = WatermarkDetector(KEY, list_size=8)
for _ in range(3): det.verify(audio, 48000)
torch.cuda.synchronize()
t0 = time.perf_counter(); r = det.verify(audio, 48000); torch.cuda.synchronize(); print(f"verify: {(time.perf_counter() - t0) * 1e3:.2f} ms -> {r}")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): det.verify(audio, 48000)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
