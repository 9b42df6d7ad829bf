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
