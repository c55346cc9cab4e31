"""The drop-in boundary is a C ABI: a host written in plain C (no Python, no torch, no C++) drives the library through include/tfusion.h
alone.  The CPU test builds that program with gcc (header is valid C99, every symbol it uses links); the GPU test runs it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_consumer", "c_abi_consumer.c")
LIBDIR = os.path.join(ROOT, "transfusion_amd", "lib")


def _build(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    from transfusion_amd import build
    build.build_lib(verbose=False)
    exe = os.path.join(str(tmp_path), "c_abi_consumer")
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", SRC,
           "-o", exe, "-L", LIBDIR, "-ltfusion_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_host_builds_against_the_header(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_plain_c_host_runs_linear_and_layernorm(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_consumer: OK" in r.stdout
