"""A/B of builds of the one-lane-per-path list decoder: python tools/wide_ab.py NAME [NAME ...]  (libechoseal_hip_NAME.so; '' = the product library).
Each build runs in its own process: SCL-L on random LLRs (B = 65 536 and 24 576 at L = 8; 8 192 at L = 32; 1 024 at L = 256), best of 3, plus a digest
of the outputs so that builds can be checked to agree bit for bit (the digest of the product build is the reference)."""
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
    eng = RxEngine(0, list_size_max=256); rng = np.random.default_rng(0)
    sigma = float(os.environ.get("ES_AB_SIGMA", "3")); zero = float(os.environ.get("ES_AB_ZERO", "0"))   # LLR scale; share of exact zeros (weak / cut frames)
    x = np.clip(rng.normal(0, sigma, (4096, 1024)), -12, 12).astype(np.float32)
    if zero: x[rng.random(x.shape) < zero] = 0.0
    base = torch.from_numpy(x).to(eng.device)
    base[::7] = torch.round(base[::7])                                   # some tie-heavy rows
    eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", 1)
    for L, B in ((8, 65536), (8, 24576), (32, 8192), (256, 1024)):
        llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
        r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        n = min(B, 4096)
        h = hashlib.sha256(r.cand_info[:n].cpu().numpy().tobytes() + r.cand_metric[:n].cpu().numpy().tobytes() + r.cand_ok[:n].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"[{name or 'product':10s}] L={L:3d} B={B:6d}: {best * 1e3:8.2f} ms -> {B / best / 1e3:8.1f} k frames/s  digest {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or [""]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], check=False)
