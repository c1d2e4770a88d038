"""The hand-written compare-exchange steps of the list decoder's sort network (inline assembly in
echoseal_amd/csrc/es_scl_wide.hip, restated in tools/ub/ub_sortce.hip) against plain C++: every step form (DPP quad_perm / row_ror
operands, fetched partner, a lane's own pair), 2 000 random trials with exact ties, +inf fillers and keys that differ in the low word only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
def test_sort_compare_exchange_steps(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    # the step macros of the unit check are the kernel's own text
    src = open(os.path.join(ROOT, "tools", "ub", "ub_sortce.hip")).read()
    ker = open(os.path.join(ROOT, "echoseal_amd", "csrc", "es_scl_wide.hip")).read()
    for needle in ('"v_sub_co_u32_dpp %3, vcc, %2, %2 " CTRL', '"s_xor_b64 vcc, vcc, %4\\n\\t"', '"v_subb_co_u32 %3, vcc, %1, %5, vcc\\n\\t"',
                   '"v_cndmask_b32 %9, %9, %6, vcc"'):
        assert needle in src and needle in ker, needle
    exe = str(tmp_path / "ub_sortce")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tools", "ub", "ub_sortce.hip")], check=True,
                   capture_output=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches: 0 " in p.stdout, p.stdout[-500:]
