"""Hardware behaviour the kernels rely on, checked on the device itself (tools/micro/*.hip, built with hipcc on the spot)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_lds_dma_lane_layout(tmp_path):
    """sdp_process fetches the next point's sub-problem descriptors with global_load_lds_dwordx4 (lra_amd/csrc/sdp.hip, NODE_FETCH): a lane's 16 bytes must land at
    the LDS base + 16 * lane, and lanes that are switched off must write nothing (the buffers hold 36 slots, the lanes beyond never have a visit)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "lds_dma")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "micro", "lds_dma.hip"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "lds_dma: ok" in r.stdout, r.stdout + r.stderr
