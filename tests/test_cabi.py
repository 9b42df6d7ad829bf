"""The C-ABI shared library: loads, exports every symbol include/mifft.h declares, validates
arguments on the host, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import hackathon_fft_amd as mf
from hackathon_fft_amd import _lib
from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mifft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mifft_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mifft.h but not exported"
    assert sorted(_lib.EXPORTS) == declared


def test_version_and_strings():
    L = _lib.lib()
    assert L.mifft_version() == 1
    assert L.mifft_status_string(-5) == b"bases do not factor the length"


def _create(device, dims, batch=4, in_dtype=0, out_dtype=0, comps=2, bases=None):
    L = _lib.lib()
    h = ctypes.c_void_p()
    c_dims = (ctypes.c_int64 * len(dims))(*dims)
    if bases is None:
        flat, lens = None, None
    else:
        f = [b for bs in bases for b in bs]
        flat = (ctypes.c_uint32 * max(1, len(f)))(*f)
        lens = (ctypes.c_int32 * len(dims))(*[len(bs) for bs in bases])
    rc = L.mifft_plan_create(ctypes.byref(h), device, in_dtype, out_dtype, len(dims), c_dims, batch, comps, 0,
                             flat, lens, 0)
    return rc, h


@pytest.mark.parametrize("kwargs,status", [
    (dict(dims=[]), -1),
    (dict(dims=[4, 4, 4, 4, 4, 4, 4]), -1),
    (dict(dims=[1]), -2),
    (dict(dims=[8, 1]), -2),
    (dict(dims=[8], comps=3), -3),
    (dict(dims=[8], out_dtype=2), -4),
    (dict(dims=[8], in_dtype=9), -4),
    (dict(dims=[8], batch=-1), -8),
    (dict(dims=[32], bases=[[4, 2]]), -5),
    (dict(dims=[8], bases=[[1, 8]]), -6),
    (dict(dims=[8], bases=[[]]), -7),
    (dict(dims=[2 * 101]), -5),
    # 32-bit lane offsets of the strided tiles: a strided dimension must span fewer than 2^32 elements
    (dict(dims=[64, 16384, 16384], batch=1), -9),
    (dict(dims=[4096, 1 << 20], batch=1), -9),
])
def test_plan_validation_happens_on_the_host(kwargs, status):
    rc, h = _create(0, **kwargs)
    assert rc == status and not h.value
    assert _lib.lib().mifft_last_error()


def test_strided_span_just_below_the_32_bit_limit_is_not_rejected_by_the_guard():
    # 15 x 2^28 - 64 elements: passes the span guard and then fails only for want of a device here (or succeeds there)
    rc, h = _create(-1, [15, 16384, 16384], batch=1)
    assert rc == -10 and not h.value   # device -1: refused AFTER the host-side checks


def test_no_cpu_fallback():
    # device -1 is never valid; with no GPU in this container device 0 is refused as well
    rc, h = _create(-1, [8])
    assert rc == -10 and not h.value
    if _lib.lib().mifft_device_count() == 0:
        rc, h = _create(0, [8])
        assert rc == -10 and not h.value
        with pytest.raises(mf.MifftError) as e:
            mf.DeviceContext()
        assert e.value.status == -10


def test_python_layout_checks_mirror_reference():
    with pytest.raises(mf.MifftError):
        mf.estimate_best_bases_nd((4, 2), (4, 2))              # rank <= 2
    with pytest.raises(mf.MifftError):
        mf.estimate_best_bases_nd((4, 8, 3), (4, 8, 2))        # C_in not in {1,2}
    with pytest.raises(mf.MifftError):
        mf.estimate_best_bases_nd((4, 8, 2), (4, 8, 1))        # C_out != 2
    with pytest.raises(mf.MifftError):
        mf.estimate_best_bases_nd((4, 8, 2), (4, 9, 2))        # leading dims differ
    with pytest.raises(mf.MifftError):
        mf.estimate_best_bases_nd((4, 1, 8, 2), (4, 1, 8, 2))  # inner dim of size 1
    assert mf.estimate_best_bases_nd((10, 640, 480, 2), (10, 640, 480, 2)) == [
        [5, 2, 2, 2, 2, 2, 2, 2], [5, 3, 2, 2, 2, 2, 2]]


def test_jit_precompile_needs_no_device():
    """hipRTC builds the runtime-specialised kernel on a machine without a GPU (the code object is only loaded at
    plan creation); a prime factor above 4093 has no fused configuration."""
    import ctypes
    from hackathon_fft_amd import _lib
    L = _lib.lib()
    sz = ctypes.c_size_t(0)
    assert L.mifft_jit_precompile(0, 0, 49, 0, 0, ctypes.byref(sz)) == 0
    assert sz.value > 4096
    assert L.mifft_jit_precompile(0, 0, 77, 1, 0, ctypes.byref(sz)) == 0   # strided form
    assert L.mifft_jit_precompile(1, 1, 121, 0, 0, ctypes.byref(sz)) == 0  # fp64
    assert L.mifft_jit_precompile(2, 0, 480, 0, 1, ctypes.byref(sz)) == 0  # uint8 real input widened in the load
    assert L.mifft_jit_precompile(3, 0, 96, 0, 0, ctypes.byref(sz)) == 0   # int32 complex input
    for code in (4, 5, 6, 7, 8):                                           # int8 / int16 / uint16 / half / bfloat16
        assert L.mifft_jit_precompile(code, 0, 96, 0, code % 2, ctypes.byref(sz)) == 0, code
    assert L.mifft_jit_precompile(9, 0, 96, 0, 0, ctypes.byref(sz)) == -4   # no such element type
    assert L.mifft_jit_precompile(0, 1, 93, 0, 0, ctypes.byref(sz)) == 0   # float input under a double plan
    assert L.mifft_jit_precompile(0, 0, 97, 0, 0, ctypes.byref(sz)) == 0    # one prime factor <= 4093: cooperative pass 0
    assert L.mifft_jit_precompile(0, 0, 4099, 0, 0, ctypes.byref(sz)) == -9
    assert L.mifft_jit_precompile(0, 0, 97, 1, 0, ctypes.byref(sz)) == 0    # ... strided too (tile staged in LDS)
    assert L.mifft_jit_precompile(0, 0, 37 * 41, 0, 0, ctypes.byref(sz)) == 0    # two large prime factors: passes 0 and 1
    assert L.mifft_jit_precompile(0, 0, 37 * 41 * 43, 0, 0, ctypes.byref(sz)) == -9   # three
    assert b"fused" in L.mifft_last_error()
    assert L.mifft_jit_precompile(0, 5, 49, 0, 0, ctypes.byref(sz)) == -4


def test_jit_disk_cache_is_shared_between_processes(tmp_path):
    """MIFFT_JIT_CACHE_DIR: the first process compiles and stores the code object, the second one loads it."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import ctypes, time, sys; sys.path.insert(0, %r); from hackathon_fft_amd import _lib; L = _lib.lib(); "
            "sz = ctypes.c_size_t(); t = time.time(); rc = L.mifft_jit_precompile(0, 0, 363, 0, 0, ctypes.byref(sz)); "
            "print(rc, sz.value, time.time() - t)" % ROOT)
    env = dict(os.environ, MIFFT_JIT_CACHE_DIR=str(tmp_path))
    first = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    files = list(tmp_path.glob("mifft_*.co"))
    assert first[0] == "0" and len(files) == 1 and files[0].stat().st_size > 4096
    second = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert second[:2] == first[:2]
    assert float(second[2]) < 0.5 * float(first[2])      # loaded, not compiled


def test_slab_plan_rejects_a_whole_batch_smaller_than_the_slab():
    L = _lib.lib()
    h = ctypes.c_void_p()
    dims = (ctypes.c_int64 * 1)(64)
    rc = L.mifft_plan_create_slab(ctypes.byref(h), 0, 0, 0, 1, dims, 8, 2, 0, None, None, 0, 4)
    assert rc == -9 or rc < 0 and "whole_batch" in L.mifft_last_error().decode()
    assert not h.value


def test_only_the_c_abi_is_exported_and_the_product_library_has_no_lab_switches():
    """libmifft.so is built with -fvisibility=hidden: the dynamic symbol table holds the entry points of include/mifft.h
    and no internal C++ function (an exported `mifft::Plan::~Plan` once interposed with a host program's class of the same
    name).  The measurement switches and the fault injection exist in libmifft_lab.so only."""
    import re
    import shutil
    import subprocess
    from hackathon_fft_amd._lib import EXPORTS
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("no nm")
    libdir = os.path.join(ROOT, "hackathon_fft_amd", "csrc")
    out = subprocess.run([nm, "-D", "--defined-only", os.path.join(libdir, "libmifft.so")], capture_output=True, text=True,
                         check=True).stdout
    funcs = [ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "Tt"]
    assert sorted(funcs) == sorted(EXPORTS), sorted(set(funcs) ^ set(EXPORTS))
    blob = open(os.path.join(libdir, "libmifft.so"), "rb").read()
    lab_switches = ["MIFFT_ND_CACHE", "MIFFT_NTS_MIN_BYTES", "MIFFT_FOURSTEP_STRIDED", "MIFFT_FOURSTEP_MIN_N", "MIFFT_DPP",
                    "MIFFT_TEST_FAIL_SCRATCH_ALLOC", "MIFFT_RADER_MIN", "MIFFT_GRID_PER_CU", "MIFFT_JIT_IMAGE", "MIFFT_ROW2D",
                    "MIFFT_JIT_NT", "MIFFT_JIT_DEFINES", "MIFFT_FS_N1"]
    # (as C strings: the kernel headers embedded for the runtime compiler mention macro names, never NUL-terminated)
    assert [n for n in lab_switches if n.encode() + b"\0" in blob] == []
    for n in ("MIFFT_JIT", "MIFFT_JIT_CACHE_DIR", "MIFFT_JIT_VERBOSE"):
        assert n.encode() + b"\0" in blob, n
    lab = os.path.join(libdir, "libmifft_lab.so")
    if os.path.exists(lab):
        lab_blob = open(lab, "rb").read()
        assert [n for n in lab_switches if n.encode() + b"\0" not in lab_blob] == []
