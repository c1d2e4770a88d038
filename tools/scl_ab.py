"""A/B of list-decoder builds: python tools/scl_ab.py NAME [NAME ...]  (libechoseal_hip_NAME.so; '' = the product library).
Each build runs in its own process: SCL-8 on random LLRs at several batch sizes (multi-frame kernel forced where
the batch allows), plus a digest of the outputs so that builds can be checked to agree bit for bit."""
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
    eng = RxEngine(0, list_size_max=16); rng = np.random.default_rng(0)
    base = torch.from_numpy(np.clip(rng.normal(0, 3, (4096, 1024)), -12, 12).astype(np.float32)).to(eng.device)
    for multi in (1, 2, 0):                       # 1: four lanes per path, 2: two lanes per path, 0: one frame per wave
        eng.set_option("scl_multi", 1 if multi else 0)
        if multi:
            try: eng.set_option("scl_lanes", 4 if multi == 1 else 2)
            except Exception: continue
        for B in (1024, 4096, 16384, 65536):
            llr = base.repeat(-(-B // 4096), 1)[:B].contiguous()
            r = eng.scl(llr, list_size=8, skip_if_hard_ok=False); torch.cuda.synchronize()
            t0 = time.perf_counter(); r = eng.scl(llr, list_size=8, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            h = hashlib.sha256(r.cand_info[:4096].cpu().numpy().tobytes() + r.cand_metric[:4096].cpu().numpy().tobytes()).hexdigest()[:12]
            print(f"[{name or 'product'}] multi={multi} L=8 B={B:6d}: {dt * 1e3:8.2f} ms -> {B / dt / 1e3:8.1f} k frames/s  digest {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or [""]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], check=False)
