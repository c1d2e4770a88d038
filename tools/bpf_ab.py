"""A/B of band-pass builds on long records: python tools/bpf_ab.py NAME [NAME ...]  ('' = the product library): 4 records x 240 000 samples
(verify() on a 5 s clip) and 1 024 frames of 1 215; a digest of the float64 output bits so that builds can be checked to agree."""
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
    eng = RxEngine(0, list_size_max=0); rng = np.random.default_rng(0)
    for B, T in ((4, 240000), (1024, 1215), (4096, 1215)):
        x = torch.from_numpy((0.1 * rng.standard_normal((B, T))).astype(np.float32)).to(eng.device)
        x[0, ::3] = 0.0; x[1 % B, :] = -0.0
        band = (torch.arange(B, device=eng.device) % 4).to(torch.uint8)
        y = eng.bpf(x, band); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); y = eng.bpf(x, band); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"[{name or 'product':10s}] B={B:5d} T={T:6d}: {best * 1e3:8.3f} ms   digest {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or [""]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], check=False)
