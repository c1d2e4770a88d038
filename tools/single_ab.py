"""A/B of builds of the one-frame-per-wave list decoder (lowest latency): python tools/single_ab.py NAME [NAME ...]  ('' = the product library)."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import echoseal_amd._native as nat
    name = sys.argv[2]
    if name:
        nat.LIB_PATH = os.path.join(os.path.dirname(nat.LIB_PATH), f"libechoseal_hip_{name}.so")
    from echoseal_amd.engine import RxEngine
    eng = RxEngine(0, list_size_max=32); rng = np.random.default_rng(0)
    base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
    base[::7] = torch.round(base[::7])
    eng.set_option("scl_multi", 0)
    for L, B in ((8, 64), (8, 1024), (8, 2048), (4, 1024), (1, 1024), (16, 1024), (32, 512)):
        llr = base[:B].contiguous()
        r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        h = hashlib.sha256(r.cand_info.cpu().numpy().tobytes() + r.cand_metric.cpu().numpy().tobytes() + r.cand_ok.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"[{name or 'product':8s}] L={L:2d} B={B:5d}: {best * 1e3:7.3f} ms  digest {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or [""]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], check=False)
