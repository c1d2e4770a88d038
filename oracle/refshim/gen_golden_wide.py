#!/usr/bin/env python3
"""Golden vectors for LARGE list sizes (64, 256 = the detector's default) from the reference.

    python -m oracle.refshim.gen_golden_wide       # writes tests/golden/polar_wide_{default,glibc}.npz

Same capture method as gen_golden.py (reference PolarCode.decode run as is; final list captured
through a module-level `sorted` spy); separate file because one L=256 decode takes the reference
about 20 s.
"""
import os
import subprocess
import sys

from oracle.refshim import gen_golden as G


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode == "all":
        env = dict(os.environ, PYTHONPATH=G.ROOT)
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_wide", "default"], cwd=G.ROOT, env=env)
        env["NPY_DISABLE_CPU_FEATURES"] = G.AVX512_OFF
        subprocess.check_call([sys.executable, "-m", "oracle.refshim.gen_golden_wide", "glibc"], cwd=G.ROOT, env=env)
        return
    import numpy as np
    from oracle.refshim.shim import load_reference
    load_reference()
    import rtwm.fastpolar as fp
    base = np.load(os.path.join(G.GOLD, "polar_default.npz"))
    cases = {k: base[f"{k}/llr"] for k in ("garbage_rng7", "det_clean_ctr0", "awgn090")}
    G.LISTS = (64, 256)
    out = G.run_polar(np, fp, cases)
    out["meta/mode"] = np.array(mode)
    np.savez_compressed(os.path.join(G.GOLD, f"polar_wide_{mode}.npz"), **out)


if __name__ == "__main__":
    main()
