"""CPU tier: the Hilbert sort key of the pruned sweeps at three / four key dimensions (csrc/kde_kernels.hpp: hilbert_key) is a bijection whose
consecutive keys are neighbouring cells - checked exhaustively on small grids by a host program built with hipcc (no GPU needed: host code only).
Nothing numeric depends on the keys (they only decide which rows share a tile); a wrong curve would cost speed, silently - hence this test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hilbert_key_is_a_space_filling_bijection(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "hilbert_check"
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", str(exe), os.path.join(ROOT, "tests", "c", "hilbert_check.cpp")],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("bijective") == 5
