"""Where do the milliseconds of a 20-step burst go?  (host enqueue time, start / end of each group's list-decoder launch)"""
import os, sys, time, numpy as np, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import DecodePipeline, RxEngine
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = RxEngine(0, list_size_max=16)
frames, _ = eng.synthetic_frames(KEY, 0, 1024)
sec = WatermarkEmbedder(KEY).sec
pn, band = eng.schedule(sec._prng.sub_key, KEY, ctr0=0, n=1024)
pipe = DecodePipeline(eng, list_size=8, lanes=4, scl_streams=2, group=G)
for _ in range(3 * G):
    pipe.submit(frames, band, pn)
pipe.synchronize(); torch.cuda.synchronize()
flushes = []
if len(sys.argv) > 3 and sys.argv[3] == "force": pipe._flush_lanes = 1
orig = pipe._flush
def probe(g):
    if g.done is None:
        j = g.index % len(pipe.backs)
        s = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(pipe.backs[j]):
            for ev in g.ready: pipe.backs[j].wait_event(ev)
            s.record()
        orig(g)
        e = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(pipe.backs[j]): e.record()
        flushes.append((g.count, s, e, time.perf_counter()))
    else:
        orig(g)
pipe._flush = probe
for rep in range(3):
    flushes.clear()
    torch.cuda.synchronize()
    t0e = torch.cuda.Event(enable_timing=True); t0e.record()
    t0 = time.perf_counter()
    for k in range(K):
        pipe.submit(frames, band, pn)
    t1 = time.perf_counter()
    pipe.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms -> {K * 1024 / (t2 - t0) / 1e6:.3f} M frames/s")
    for cnt, s, e, th in flushes:
        print(f"   group of {cnt}: flushed by the host at {1e3 * (th - t0):.2f} ms; decoder start {t0e.elapsed_time(s):.2f} ms, end {t0e.elapsed_time(e):.2f} ms")
