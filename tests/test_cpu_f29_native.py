"""Host differential test of the MSM kernels' number format (fq29.cuh / ec29.cuh: 9 x 29-bit lazy
Montgomery field and XYZZ formulas with per-site K*p offsets) against the canonical 8 x 32-bit
code, with the value/limb bound assertions compiled in (-DG16_F29_CHECK).  The canonical code is
itself pinned to the Python oracle by tests/test_cpu_host.py (setup tool == oracle, byte for byte)."""
import os
import subprocess

from conftest import ROOT


def test_f29_field_and_curve_differential(tmp_path):
    exe = tmp_path / "f29_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DG16_F29_CHECK",
                           "-I", os.path.join(ROOT, "nzcp-circom_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "f29_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout


def test_f29_bounds_script_closes():
    """tools/f29_bounds.py: the K constants of ec29.cuh keep every product input below 16p."""
    out = subprocess.run(["python3", os.path.join(ROOT, "tools", "f29_bounds.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "G1 BX=" in out.stdout and "G2 BX=" in out.stdout
