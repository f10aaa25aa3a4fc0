"""CPU: the table-driven log / sincos of the device's coefficient phase (mcmc_gpu_amd/csrc/math_tables.h compiles for the
host too) against long double arithmetic: log within 3e-16 + 2.5e-16 |log x| (uniforms, uniforms next to 1, general
positive arguments), sqrt(-2 log u) within 1.5e-15, sincos within 3e-16 absolute."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_log_and_sincos_tables_against_long_double(tmp_path):
    exe = tmp_path / "math_tables_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", str(ROOT / "mcmc_gpu_amd" / "csrc"), "-o", str(exe),
                    str(ROOT / "tests" / "native" / "math_tables_check.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
