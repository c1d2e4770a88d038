import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from echoseal_amd.engine import RxEngine
eng = RxEngine(0, list_size_max=64); rng = np.random.default_rng(0)
base = torch.from_numpy(np.clip(rng.normal(0, 3, (16384, 1024)), -12, 12).astype(np.float32)).to(eng.device)
base[::9] = torch.clamp(base[::9] * 6, -12, 12)
L, B = int(sys.argv[1]), int(sys.argv[2])
llr = base.repeat(-(-B // 16384), 1)[:B].contiguous()
out = {}
for lanes in (0, 1):
    eng.set_option("scl_multi", 1); eng.set_option("scl_lanes", lanes)
    out[lanes] = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
for k in ("hard_info", "hard_ok", "cand_info", "cand_metric", "cand_ok", "ncand"):
    a, b = getattr(out[0], k), getattr(out[1], k)
    d = (a != b).reshape(B, -1).any(dim=1).nonzero().flatten().cpu().numpy()
    print(k, "frames differing:", len(d), d[:20], d[-5:] if len(d) else "")
# which one is right? rows repeat every 16384
for lanes in (0, 1):
    m = out[lanes].cand_metric
    rep = (m[:16384] != m[16384:32768]).any(dim=1).sum().item() if B >= 32768 else -1
    print("lanes", lanes, "rows whose repeat differs from the first copy:", rep)
for lanes in (0, 1):
    eng.set_option("scl_lanes", lanes)
    for rep in range(3):
        r = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        m = r.cand_metric.reshape(-1, 16384, r.cand_metric.shape[-1]) if B % 16384 == 0 else None
        bad = (m != m[0:1]).any(dim=2).nonzero().cpu().numpy() if m is not None else []
        print("lanes", lanes, "run", rep, "rows differing from their first copy:", len(bad), [(int(a) * 16384 + int(b)) for a, b in bad[:12]])
