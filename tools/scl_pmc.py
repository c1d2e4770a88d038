import sys, numpy as np, torch
sys.path.insert(0, '.')
from echoseal_amd.engine import RxEngine
eng = RxEngine(0)
rng = np.random.default_rng(0)
llr = torch.from_numpy(np.clip(rng.normal(0, 3, (1024, 1024)), -12, 12).astype(np.float32)).to(eng.device)
for _ in range(3):
    eng.scl(llr, list_size=8)
torch.cuda.synchronize()
