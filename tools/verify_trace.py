"""One verify() of a 5 s clip (list size 8) for rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder
from echoseal_amd.detector import WatermarkDetector
KEY = b"\xAA" * 32
rng = np.random.default_rng(1)
audio = WatermarkEmbedder(KEY).process((0.05 * rng.standard_normal(5 * 48000)).astype(np.float32))
det = WatermarkDetector(KEY, list_size=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
det.verify(audio[:48000], 48000); torch.cuda.synchronize()
for _ in range(3):
    det.verify(audio, 48000)
torch.cuda.synchronize()
