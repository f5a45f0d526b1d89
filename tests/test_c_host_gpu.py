"""A non-Python host of the drop-in boundary: tools/c_host_demo.c is plain C11 over include/unitex_hip.h, built with gcc (not hipcc), linked against
libunitex_hip.so and the HIP runtime only.  It assembles a small FLUX-shaped step with utx_dit_load from raw hipMalloc pointers, runs it twice through
utx_dit_step and verifies a finite, non-trivial, reproducible prediction.  (That the C-built plan equals the Python-built one entry for entry, and the oracle
numerically, is tests/test_dit_ops_gpu.py's job.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_host_builds_and_runs_a_dit_step(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    lib_dir = os.path.join(ROOT, "unitex_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libunitex_hip.so")), "build the library first (python __graft_entry__.py build)"
    exe = str(tmp_path / "c_host_demo")
    cmd = [gcc, "-O2", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "c_host_demo.c"), "-L" + lib_dir, "-lunitex_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir,
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "finite 1 reproducible 1" in r.stdout, r.stdout
