"""Can a whole decode_batch (11 kernel launches over two streams) be captured into a HIP graph and replayed?
Prints per-batch latency eager vs replayed for a small batch, and checks the replayed results."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
from echoseal_amd.engine import RxEngine
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8); dev = eng.device
for B in (64, 1024):
    tx = WatermarkEmbedder(KEY); ctrs = list(range(B))
    frames = torch.from_numpy(tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))).to(dev)
    band = torch.from_numpy(np.array([band_index(KEY, c) for c in ctrs], np.uint8)).to(dev)
    pn = torch.from_numpy(tx.sec.pn_bytes_batch(ctrs, 152)).to(dev)
    for _ in range(3):
        ref = eng.decode_batch(frames, band, pn, list_size=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.decode_batch(frames, band, pn, list_size=8)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(2):
            eng.decode_batch(frames, band, pn, list_size=8)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = eng.decode_batch(frames, band, pn, list_size=8)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 20
    same = all(torch.equal(a, b) for a, b in ((out[0].thr, ref[0].thr), (out[0].peaks, ref[0].peaks), (out[1], ref[1]),
                                              (out[2].cand_info, ref[2].cand_info), (out[2].cand_metric, ref[2].cand_metric)))
    print(f"B={B}: eager {eager * 1e3:.3f} ms, graph replay {rep * 1e3:.3f} ms per batch; results identical: {same}", flush=True)
