"""Does the interpreter exit cleanly after using the engine in various ways?  python tools/exit_probe.py  (each mode in a child process; prints its exit code)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ["front", "front_noweak", "front_sel", "front_keep", "engine", "lanes", "grouped"]
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from echoseal_amd.engine import RxEngine, DecodePipeline, pipeline_streams
    mode = sys.argv[2]
    eng = RxEngine(0, list_size_max=8)
    if mode == "front_engine":
        e2 = RxEngine(0, list_size_max=0)
    if mode == "streams":
        st = pipeline_streams(eng.device, 4)
    if mode.startswith("front"):
        import echoseal_amd.engine as E
        if mode == "front_noweak":
            class _S(set):
                def add(self, x): pass
            E._LIVE_STREAMS = _S()
        from echoseal_amd import workloads as WL
        frames, band, pn, _ = WL.c2_frames(range(256))
        f, b, p = (torch.from_numpy(x).to(eng.device) for x in (frames, band, pn))
        pipe = DecodePipeline(eng, list_size=8)
        out = [pipe.submit(f, b, p) for _ in range(7)]
        pipe.synchronize()
        if mode == "front_keep":
            import builtins; builtins._keep = (pipe, out)
    if mode in ("lanes", "grouped", "grouped_nocycle"):
        from echoseal_amd import workloads as WL
        frames, band, pn, _ = WL.c2_frames(range(256))
        f, b, p = (torch.from_numpy(x).to(eng.device) for x in (frames, band, pn))
        pipe = DecodePipeline(eng, list_size=8, lanes=2, scl_streams=2, group=3) if mode != "lanes" else DecodePipeline(eng, list_size=8, lanes=3)
        out = [pipe.submit(f, b, p) for _ in range(7)]
        pipe.synchronize()
        if mode == "grouped_nocycle":
            for o in out:
                o[2].g.pipe = None
            del out, pipe
            import gc; gc.collect()
    if mode == "scl_only":
        llr = torch.randn(512, 1024, device=eng.device)
        r = eng.scl(llr, list_size=8, skip_if_hard_ok=False); r.check()
    if mode == "detector":
        from echoseal_amd.detector import WatermarkDetector
        d = WatermarkDetector(b"\xAA" * 32, list_size=8)
        d.verify(np.zeros(48000, np.float32), 48000)
    torch.cuda.synchronize()
    print("child done", mode, flush=True)
    sys.exit(0)
for m in MODES:
    p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "--child", m], capture_output=True, text=True)
    print(f"{m:16s} rc={p.returncode}  {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ''}  {p.stderr.strip().splitlines()[-3:] if p.returncode else ''}", flush=True)
