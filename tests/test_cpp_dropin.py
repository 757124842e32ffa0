"""The C++ host-side mirror (lp_mp_amd/include/LP_gpu.hxx): compile the reference-style test program with
g++ against the C ABI library; run its host-only part here and the full program on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from lp_mp_amd import build as B
    B.build()
    exe = str(tmp_path / "test_model_gpu")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "lp_mp_amd", "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_model_gpu.cpp"),
                           "-L", B.CSRC, "-llpmp_engine", "-Wl,-rpath," + B.CSRC])
    return exe


def test_cpp_mirror_compiles_and_host_part_passes(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.check_output([exe, "--host-only"], text=True)
    assert "all tests passed" in out


@pytest.mark.gpu
def test_cpp_mirror_full_run_on_device(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.check_output([exe], text=True, timeout=600)
    assert "all tests passed" in out
