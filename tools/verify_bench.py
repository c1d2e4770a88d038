"""WatermarkDetector.verify() on a 5 s clip (the f-1 / f-2 flow: ONE sync launch sequence over the four bands, one header
decode over all peaks, per band one demodulate + list-decode + validate batch -- all on the GPU): wall time per call, the
sync stage alone (the reference needs ~67 ms per band on a CPU core for it), and verify_batch on 8 clips."""
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
    det._trace = []; det.verify(audio, 48000)
    print(f"list_size={L:3d}: verify(5 s clip) -> {ok} in {dt * 1e3:.1f} ms ({len(det._trace)} (peak, counter) tries)", flush=True)
det = WatermarkDetector(KEY, list_size=8)
det._scan_prepare([audio], det._band_order()); torch.cuda.synchronize()
t0 = time.perf_counter(); det._scan_prepare([audio], det._band_order()); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sync of the four bands + header decode of every peak, 5 s clip: {dt * 1e3:.2f} ms (reference: ~67 ms per band for sync alone)")
clips = [tx.process((0.05 * rng.standard_normal(5 * 48000)).astype(np.float32)) for _ in range(8)]
det.verify_batch(clips, 48000); torch.cuda.synchronize()
t0 = time.perf_counter(); res = det.verify_batch(clips, 48000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"verify_batch(8 clips of 5 s, list 8): {dt * 1e3:.1f} ms total, {dt * 1e3 / 8:.1f} ms per clip -> {res}")
# stage split of the sync of one 5 s clip (four bands): band-pass, correlation + threshold / peaks
eng = det.engine
x = torch.from_numpy(np.repeat(audio[None, :], 4, axis=0)).to(eng.device)
bid = torch.arange(4, dtype=torch.uint8, device=eng.device)
for name, fn in (("band-pass (4 records x 240 000 samples)", lambda: eng.bpf(x, bid)), ("sync (band-pass + correlation + threshold / peaks)", lambda: eng.sync(x, bid, keep_corr=False))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.2f} ms")
