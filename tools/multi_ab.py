"""A/B of builds of the several-frames-per-wave list decoder: python tools/multi_ab.py NAME [NAME ...]  ('' = the product library)."""
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
    eng.set_option("scl_multi", 1)
    for lanes, L, B in ((4, 8, 1024), (4, 8, 16384), (4, 8, 65536), (2, 8, 65536), (4, 16, 8192), (2, 32, 4096), (4, 1, 65536)):
        eng.set_option("scl_lanes", lanes)
        llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
        r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        n = min(B, 4096)
        h = hashlib.sha256(r.cand_info[:n].cpu().numpy().tobytes() + r.cand_metric[:n].cpu().numpy().tobytes() + r.cand_ok[:n].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"[{name or 'product':8s}] lanes/path={lanes} L={L:2d} B={B:6d}: {best * 1e3:8.2f} ms -> {B / best / 1e3:8.1f} k frames/s  digest {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or [""]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], check=False)
