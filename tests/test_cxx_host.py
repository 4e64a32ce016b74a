"""The C++ host layer (include/orbslam_hip.hpp) over the C-ABI: compiles with g++
against liborbslam_hip.so; on the GPU box it runs the whole path, without a GPU
it must fail loudly with ORBX_ERR_NO_DEVICE."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "orb_slam2_e_amd")


def _build(tmp_path):
    from orb_slam2_e_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    exe = str(tmp_path / "dropin_smoke")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "dropin_smoke.cpp"), "-o", exe,
                           "-L", LIBDIR, "-lorbslam_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_host_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = _build(tmp_path)
    out = subprocess.run([exe, "nodevice"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK nodevice" in out.stdout


@pytest.mark.gpu
def test_cxx_host_runs_the_hot_path(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")
