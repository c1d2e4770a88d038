"""Host cost of the pieces of one pipelined submit (tiny batch, so the GPU never pushes back)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.engine import RxEngine, DecodePipeline
from echoseal_amd.embedder import WatermarkEmbedder
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=8)
frames, _ = eng.synthetic_frames(KEY, 0, 64)
sec = WatermarkEmbedder(KEY).sec
pn, band = eng.schedule(sec._prng.sub_key, KEY, ctr0=0, n=64)
st = torch.cuda.Stream(eng.device)
N = 2000
def t(name, fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    dt = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    print(f"{name:34s} {dt * 1e6:7.1f} us", flush=True)
def ctx():
    with torch.cuda.stream(st): pass
t("with torch.cuda.stream(st): pass", ctx)
t("st.wait_stream(current)", lambda: st.wait_stream(torch.cuda.current_stream(eng.device)))
t("Event() + record()", lambda: torch.cuda.Event().record())
t("torch.empty x7", lambda: [torch.empty((64, 1215), dtype=torch.float64, device=eng.device) for _ in range(7)])
t("record_stream x3", lambda: [x.record_stream(st) for x in (frames, band, pn)])
y, y32 = eng.bpf2(frames, band)
t("bpf2", lambda: eng.bpf2(frames, band))
t("sync_fused", lambda: eng.sync_fused(y, y32, band))
out = torch.empty((64, 1024), dtype=torch.float32, device=eng.device)
t("llr(out=)", lambda: eng.llr(y, band, pn, variant=0, out=out))
pipe = DecodePipeline(eng, list_size=8, lanes=4, scl_streams=2, group=1 << 20)
t("grouped submit (whole)", lambda: pipe.submit(frames, band, pn))
