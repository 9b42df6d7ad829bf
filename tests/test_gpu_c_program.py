"""The drop-in boundary from plain C: tests/c/cabi_demo.c includes include/mifft.h, links libmifft.so and runs
without Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_program_through_the_c_abi(tmp_path):
    gcc = shutil.which("gcc") or "gcc"
    libdir = os.path.join(ROOT, "hackathon_fft_amd", "csrc")
    exe = str(tmp_path / "cabi_demo")
    # plain gcc: the HIP runtime is needed only for hipMalloc / hipMemcpy of the test buffers
    subprocess.run([gcc, "-O2", "-std=gnu11", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "cabi_demo.c"),
                    "-L", libdir, "-lmifft", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bases do not factor the length" in r.stdout


def test_cxx_program_through_the_host_mirror(tmp_path):
    """include/mifft.hpp: the reference's plan_fft / fft call surface as a header-only C++ layer over the C ABI (the
    reference is compiled code; its Mojo toolchain is absent), driven from a plain g++ program."""
    gxx = shutil.which("g++") or "g++"
    libdir = os.path.join(ROOT, "hackathon_fft_amd", "csrc")
    exe = str(tmp_path / "cxx_demo")
    subprocess.run([gxx, "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "cxx_demo.cpp"),
                    "-L", libdir, "-lmifft", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "cxx demo ok" in r.stdout, r.stdout + r.stderr
    assert "The rank should be bigger than 2." in r.stdout and "rader" in r.stdout
