"""Build lint (no GPU): the prefill GEMM kernels keep no kernel-argument load behind their K loop and the ring kernels
stay within 168 VGPRs without spills (tools/check_kernarg_reloads.py). Needs hipcc — the kernels are
cross-compiled to gfx950 assembly, nothing runs."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_gemm_kernels_have_no_late_kernarg_loads_and_ring_kernels_fit(tmp_path):
    isa = str(tmp_path / "woq_gemm_f16.s")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernarg_reloads.py"), "--compile-to", isa],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GEMM kernels checked" in r.stdout
