"""Device-side invariants the kernels rely on, checked by the stand-alone micro programs under scripts/micro (compiled here with hipcc)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
def test_dpp_and_permlane_reductions_match_shfl_xor_bit_for_bit(tmp_path):
    """common.h lanes_reduce / rows_reduce: the DPP row steps (quad_perm, row_half_mirror, row_mirror) and the
    v_permlane16 / 32_swap steps give, lane for lane, what __shfl_xor (ds_bpermute) gives for xor 1, 2, 4, 8, 16, 32."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not on this box")
    exe = str(tmp_path / "lane_reduce")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "scripts", "micro", "lane_reduce.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all six steps bit-identical to __shfl_xor" in r.stdout
