import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_gpu_parity as tg
from echoseal_amd.engine import RxEngine
from oracle import oracle as O
eng=RxEngine(0)
orig=eng.sync_fast
def spy(f,b):
    r=orig(f,b); torch.cuda.synchronize()
    print("spy: B,T=",tuple(f.shape),"flags",torch.unique(r.flags,return_counts=True), "f dtype", f.dtype, "contig", f.is_contiguous(), "b", b.dtype, b.shape)
    r2=orig(f.clone(),b.clone()); torch.cuda.synchronize()
    print("spy clone:",torch.unique(r2.flags,return_counts=True))
    return r
eng.sync_fast=spy
try:
    tg.test_sync_fast_equals_float64_path(eng, O)
    print("direct call: PASS")
except AssertionError as e:
    print("direct call: FAIL", str(e)[:200])
