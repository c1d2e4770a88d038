"""CPU: the resampling oracle (oracle/c/eso_resample.c: the compiled inner loop of scipy.signal.upfirdn, restated) and
the host plan (echoseal_amd.utils.resample_plan: what SciPy does in Python before that loop) against
scipy.signal.resample_poly itself -- the third-party call behind the reference's resample_to (rtwm/utils.py:58-66)."""
import math

import numpy as np
import pytest
from scipy.signal import resample_poly

from echoseal_amd.utils import resample_plan


@pytest.fixture(scope="module")
def oracle():
    import oracle.oracle as o
    o.build()
    return o


@pytest.mark.parametrize("fs_in", [44100, 22050, 32000, 96000, 16000, 47999, 8000, 48000])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16])
def test_oracle_resample_equals_scipy(oracle, fs_in, dtype):
    rng = np.random.default_rng(fs_in)
    n = int(rng.integers(300, 5000))
    x = rng.normal(0, 0.3, n)
    x = (x * 20000).astype(np.int16) if dtype == np.int16 else x.astype(dtype)
    g = math.gcd(fs_in, 48000)
    up, down = 48000 // g, fs_in // g
    ref = resample_poly(x, up, down)
    got = oracle.resample_poly(x, up, down)
    assert ref.dtype == got.dtype and ref.shape == got.shape and np.array_equal(ref.view(np.uint8), got.view(np.uint8))
    plan = resample_plan(n, up, down, x.dtype)                     # the product's own plan agrees with the oracle's
    oplan = oracle.resample_plan(n, up, down, x.dtype)
    if plan is None:
        assert oplan is None
    else:
        assert np.array_equal(plan[0], oplan[0]) and tuple(plan[1:6]) == tuple(oplan[1:6]) and plan[6] == oplan[6]
