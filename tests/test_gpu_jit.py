"""Runtime-specialised tile kernels (hackathon_fft_amd/csrc/kernels_jit.cpp): lengths without a precompiled table
entry get the same fused kernel template, compiled with hipRTC at plan creation.  Checked against the oracle (small
lengths) and fp64 pocketfft, contiguous and strided, fp32 / fp64, real input, inverse round trips, ragged batches."""
import os
import numpy as np
import pytest
import torch

import hackathon_fft_amd as mf
from conftest import REL_L2_TOL_F32, REL_L2_TOL_F64, from_complex, rel_l2, to_complex
from oracle import mifft_oracle as O

pytestmark = pytest.mark.gpu


def _run(x_np, out_dtype=None, inverse=False, bases=None):
    x = torch.from_numpy(x_np).to("cuda:0")
    odt = x.dtype if out_dtype is None else out_dtype
    out = torch.full(tuple(x.shape[:-1]) + (2,), float("nan"), device="cuda:0", dtype=odt)
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(x.dtype, odt, x.shape, out.shape, bases=bases, inverse=inverse, ctx=ctx)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    return out.cpu().numpy(), plan


# 7^2, 11^2, 7^3, 7*11, primes as one pass, 11^3, 3*11*31, 2*3*5*7*11, 13^2, 17^2, 31^2, 29*31, 3^8, 23*4, 19*27
JIT_ROWS = [49, 121, 343, 77, 31, 7, 13, 1331, 1023, 2310, 169, 289, 961, 899, 6561, 92, 513]


@pytest.mark.parametrize("n", JIT_ROWS)
def test_rows_fp32(n):
    rng = np.random.default_rng(n)
    batch = 131 if n <= 512 else 9
    x = rng.standard_normal((batch, n, 2)).astype(np.float32)
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_jit"), plan.kernel_name(0)
    assert not np.isnan(out).any()
    truth = np.fft.fft(to_complex(x), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32, plan.kernel_name(0)
    if n <= 1400:   # the oracle (the reference's literal stages) in seconds
        assert rel_l2(out, O.fftn(x)) < REL_L2_TOL_F32
    back, _ = _run(out, inverse=True)
    assert rel_l2(back, x) < REL_L2_TOL_F32


@pytest.mark.parametrize("n,inner", [(49, 40), (121, 33), (77, 16), (343, 20), (31, 64), (1331, 17), (169, 48)])
def test_cols_fp32(n, inner):
    rng = np.random.default_rng(n + inner)
    x = rng.standard_normal((3, n, inner, 2)).astype(np.float32)
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_jit"), plan.kernel_name(0)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    back, _ = _run(out, inverse=True)
    assert rel_l2(back, x) < REL_L2_TOL_F32


@pytest.mark.parametrize("n", [49, 121, 343, 847, 31])
def test_rows_fp64(n):
    rng = np.random.default_rng(n + 7)
    x = rng.standard_normal((67, n, 2))
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_jit"), plan.kernel_name(0)
    assert rel_l2(out, O.fftn(x)) < REL_L2_TOL_F64
    back, _ = _run(out, inverse=True)
    assert rel_l2(back, x) < REL_L2_TOL_F64


@pytest.mark.parametrize("n", [49, 363])
def test_real_input_twin(n):
    rng = np.random.default_rng(n + 11)
    x = rng.standard_normal((45, n, 1)).astype(np.float32)
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_r_jit"), plan.kernel_name(0)
    truth = np.fft.fft(x[..., 0].astype(np.float64), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32


@pytest.mark.parametrize("shape,dtype,comps", [((5, 32, 32), np.float32, 2), ((3, 96, 96), np.float32, 2),
                                               ((2, 100, 100), np.float32, 2), ((7, 48, 48), np.float32, 1),
                                               ((2, 3, 120, 120), np.float32, 2), ((3, 64, 64), np.float64, 2),
                                               ((2, 64, 64), np.uint8, 1), ((300, 16, 16), np.float32, 2),
                                               ((260, 126, 126), np.float32, 2),
                                               # rectangular planes: two LDS twiddle tables
                                               ((5, 64, 128), np.float32, 2), ((5, 128, 64), np.float32, 2),
                                               ((3, 96, 160), np.float32, 2), ((7, 40, 48), np.float32, 1),
                                               ((2, 3, 120, 100), np.float32, 2), ((3, 32, 64), np.float64, 2),
                                               ((9, 32, 512), np.float32, 2), ((300, 24, 16), np.float32, 2)])
def test_planes_are_fused(shape, dtype, comps):
    """Two innermost dimensions whose plane fits LDS run as ONE pass (plane_kernel specialised at plan time): rows
    from HBM, columns inside LDS, one store."""
    rng = np.random.default_rng(sum(shape) + comps)
    if dtype == np.uint8:
        x = rng.integers(0, 255, size=shape + (comps,)).astype(dtype)
    else:
        x = rng.standard_normal(shape + (comps,)).astype(dtype)
    odt = torch.float64 if dtype == np.float64 else torch.float32
    out, plan = _run(x, out_dtype=odt)
    nd = len(shape) - 1
    assert plan.kernel_name(nd - 1).startswith("plane") and plan.kernel_name(nd - 2) == plan.kernel_name(nd - 1)
    assert plan.num_launches == nd - 1
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    tol = REL_L2_TOL_F64 if dtype == np.float64 else REL_L2_TOL_F32
    assert rel_l2(out, from_complex(truth, np.float64)) < tol
    if comps == 2 and dtype != np.uint8:
        back, _ = _run(out, inverse=True)
        assert rel_l2(back, x) < tol


@pytest.mark.parametrize("n,dtype", [(4984, np.float64), (6000, np.float64), (7168, np.float64), (10000, np.float32),
                                     (15625, np.float32), (12000, np.float32)])
def test_long_rows_one_per_workgroup(n, dtype):
    """Rows of up to 128 KiB (16384 points fp32, 8192 fp64) without a table entry: one row per workgroup, twiddles
    from the global table when the LDS table no longer fits."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((5, n, 2)).astype(dtype)
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_jit") and plan.num_launches == 1, plan.kernel_name(0)
    truth = np.fft.fft(to_complex(x), axis=1)
    tol = REL_L2_TOL_F32 if dtype == np.float32 else 1e-11
    assert rel_l2(out, from_complex(truth, np.float64)) < tol
    back, _ = _run(out, inverse=True)
    assert rel_l2(back, x) < tol


def test_mixed_nd_with_a_jit_dimension():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 49, 12, 121, 2)).astype(np.float32)
    out, plan = _run(x)
    names = [plan.kernel_name(d) for d in range(3)]
    assert names[0].endswith("_jit") and names[2].endswith("_jit"), names
    truth = np.fft.fftn(to_complex(x), axes=(1, 2, 3))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32


def test_results_do_not_depend_on_the_tile_slot_jit():
    rng = np.random.default_rng(23)
    one = rng.standard_normal((1, 343, 2)).astype(np.float32)
    x = np.repeat(one, 70, axis=0)
    out, _ = _run(x)
    for i in range(1, 70):
        assert np.array_equal(out[i], out[0]), i


# lengths with ONE prime factor in (31, 4093]: that factor is pass 0, run in LDS (TileCfg::BIGP0) -- as Rader's cyclic
# convolution when R - 1 splits into register butterflies, as the cooperative conjugate-pair pass otherwise
BIG_PRIME_ROWS = [97, 37, 74, 123, 127, 194, 101, 113, 122, 555, 328, 89 * 12, 4 * 9 * 43, 131, 251, 509, 1009, 1021,
                  2 * 251, 3 * 337, 2039, 4093,
                  37 * 41, 41 * 53, 59 * 73, 2 * 37 * 41, 3 * 43 * 47]   # two primes above 31: passes 0 and 1


@pytest.mark.parametrize("n", BIG_PRIME_ROWS)
def test_rows_with_a_large_prime_factor(n):
    rng = np.random.default_rng(n)
    batch = 131 if n <= 512 else 9
    x = rng.standard_normal((batch, n, 2)).astype(np.float32)
    bases = None
    try:
        out, plan = _run(x)
    except mf.MifftError:   # the reference's default radix estimate stops at 97 / trial division by 2..32: pass bases
        f, m, d = [], n, 2
        while m > 1:
            while m % d == 0:
                f.append(d)
                m //= d
            d += 1
        bases = [sorted(set(f))]
        out, plan = _run(x, bases=bases)
    assert plan.kernel_name(0).endswith("_jit"), plan.kernel_name(0)
    assert not np.isnan(out).any()
    truth = np.fft.fft(to_complex(x), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32, plan.kernel_name(0)
    if n <= 600:
        assert rel_l2(out, O.fftn(x, bases=bases)) < REL_L2_TOL_F32
    back, _ = _run(out, inverse=True, bases=bases)
    assert rel_l2(back, x) < REL_L2_TOL_F32


@pytest.mark.parametrize("n,rader", [(97, True), (37, True), (194, True), (1009, True), (4093, True), (2 * 251, True),
                                     (83, False),         # 82 = 2 * 41: no in-place split, below the padded threshold
                                     (509, True),         # 508 = 4 * 127: zero-padded convolution of length 1024
                                     (2039, True), (263, True), (3 * 337, True),   # ... 4096; 262 = 2 * 131 -> 528; 336: in place
                                     (167, False),        # 166 = 2 * 83, below the padded threshold of 200
                                     (37 * 41, False),    # two such primes below 128: both cooperative
                                     (131 * 37, True)])   # ... 131 by Rader, 37 cooperative
def test_which_prime_passes_run_as_rader_convolutions(n, rader):
    """Prime radices above 32: x[g^q] (*) W_R^(g^-q) through an (R - 1)-point FFT, a pointwise product with the
    precomputed spectrum and an inverse FFT, all inside the LDS tile (tile_kernel.h rader_pass) -- 2x faster at R = 97,
    6x at 1009, 14x at 4093 than the O(R^2 / 4) cooperative pass.  Primes whose R - 1 has a prime factor above 31 embed the
    convolution in a zero-padded one of smooth length (scratch block in LDS) from R = 200 on; below that the cooperative
    pass remains.  Parity against fp64 pocketfft, forward and inverse, complex, real and fp64."""
    rng = np.random.default_rng(n)
    f, m, d = [], n, 2
    while m > 1:
        if m % d == 0:
            f.append(d)
            while m % d == 0:
                m //= d
        d += 1
    for dtype, comps, tol in ((np.float32, 2, REL_L2_TOL_F32), (np.float32, 1, REL_L2_TOL_F32), (np.float64, 2, 1e-11)):
        x = rng.standard_normal((7, n, comps)).astype(dtype)
        out, plan = _run(x, bases=[f])
        # (fp64 at 4093 points: the tables -- 2 x 4092 complex doubles -- no longer fit LDS beside the row: cooperative)
        want = rader and not (n in (4093, 2039) and dtype == np.float64)
        assert ("_rader" in plan.kernel_name(0)) == want, plan.kernel_name(0)
        z = x[..., 0].astype(np.float64) + (1j * x[..., 1] if comps == 2 else 0)
        assert rel_l2(out, from_complex(np.fft.fft(z, axis=1), np.float64)) < tol, plan.kernel_name(0)
        if comps == 2:
            back, _ = _run(out, inverse=True, bases=[f])
            assert rel_l2(back, x) < tol


def test_large_prime_factor_fp64_and_real_and_uint8():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((40, 97, 2))
    out, plan = _run(x)
    assert plan.kernel_name(0).endswith("_jit")
    assert rel_l2(out, O.fftn(x)) < REL_L2_TOL_F64
    xr = rng.standard_normal((33, 74, 1)).astype(np.float32)
    out, plan = _run(xr)
    assert plan.kernel_name(0).endswith("_r_jit"), plan.kernel_name(0)
    assert rel_l2(out, from_complex(np.fft.fft(xr[..., 0].astype(np.float64), axis=1), np.float64)) < REL_L2_TOL_F32
    x8 = rng.integers(0, 255, size=(21, 97, 2)).astype(np.uint8)
    out, plan = _run(x8, out_dtype=torch.float32)
    assert "_u8" in plan.kernel_name(0)
    assert rel_l2(out, from_complex(np.fft.fft(to_complex(x8.astype(np.float64)), axis=1), np.float64)) < REL_L2_TOL_F32


def test_lengths_outside_the_jit_stay_on_the_literal_stages():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 4099, 2)).astype(np.float32)    # prime > 4093
    out, plan = _run(x, bases=[[4099]])
    assert plan.kernel_name(0) == "generic"
    assert rel_l2(out, from_complex(np.fft.fft(to_complex(x), axis=1), np.float64)) < REL_L2_TOL_F32
    x = rng.standard_normal((3, 37 * 41 * 5, 4, 2)).astype(np.float32)  # strided, two prime factors > 31, more passes
    out, plan = _run(x, bases=[[41, 37, 5], [2]])
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32


@pytest.mark.parametrize("n,inner,dtype", [(97, 20, np.float32), (74, 40, np.float32), (123, 16, np.float64),
                                           (37 * 41, 24, np.float32), (2 * 37 * 43, 16, np.float32),
                                           (555, 33, np.float32), (37, 64, np.float32), (101, 24, np.float32)])
def test_strided_dimension_with_a_large_prime_factor(n, inner, dtype):
    """Column tiles stage their tile in LDS when pass 0 is the cooperative prime pass; a prime length also stores from
    LDS (runs of TILE adjacent columns both ways)."""
    rng = np.random.default_rng(n + inner)
    x = rng.standard_normal((3, n, inner, 2)).astype(dtype)
    bases = None if n != 101 else [[101], [2, 3]]
    out, plan = _run(x, bases=bases)
    assert plan.kernel_name(0).startswith("cols") and plan.kernel_name(0).endswith("_jit"), plan.kernel_name(0)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    tol = REL_L2_TOL_F32 if dtype == np.float32 else REL_L2_TOL_F64
    assert rel_l2(out, from_complex(truth, np.float64)) < tol
    back, _ = _run(out, inverse=True, bases=bases)
    assert rel_l2(back, x) < tol


@pytest.mark.parametrize("in_dtype,comps", [(np.uint8, 1), (np.uint8, 2), (np.int32, 1), (np.int32, 2)])
@pytest.mark.parametrize("shape", [(9, 1024), (5, 49), (3, 40, 48), (2, 1080, 1920)])
def test_integer_input_is_widened_in_the_first_pass(in_dtype, comps, shape):
    """uint8 / int32 tensors (images) run on the fused kernels: the pass that reads x is specialised for the input
    element type (the reference casts in its load, fft/fft/_fft.mojo:243-251); later passes are the table kernels."""
    rng = np.random.default_rng(sum(shape) + comps)
    hi = 255 if in_dtype == np.uint8 else 100000
    x = rng.integers(0, hi, size=shape + (comps,)).astype(in_dtype)
    out, plan = _run(x, out_dtype=torch.float32)
    last = plan.kernel_name(len(shape) - 2)
    assert last.endswith("_jit") and ("_u8" in last or "_i32" in last), last
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1] if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    if np.prod(shape) <= 20000:
        assert rel_l2(out, O.fftn(x, out_dtype=np.float32)) < REL_L2_TOL_F32


def test_float_input_under_a_double_plan():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 480, 2)).astype(np.float32)
    out, plan = _run(x, out_dtype=torch.float64)
    assert plan.kernel_name(0).endswith("_f32in_jit"), plan.kernel_name(0)
    truth = np.fft.fft(to_complex(x.astype(np.float64)), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F64


_UNDER_PROFILER = r"""
import sys, torch
sys.path.insert(0, %(root)r)
import hackathon_fft_amd as mf
for shape in [(8, 343), (1, 7680, 64)]:
    full = tuple(shape) + (2,)
    plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=mf.DeviceContext(0))
    print("KERNELS", shape, [plan.kernel_name(d) for d in range(len(shape) - 1)])
"""


def test_runtime_specialisation_also_works_under_the_profiler(tmp_path):
    # hipRTC has no <hip/hip_runtime.h> on its search path inside a rocprofv3 run: the shared headers must not ask for it,
    # or every specialised plan silently degrades to the literal-stage kernels exactly when it is being measured
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    prof = shutil.which("rocprofv3")
    if prof is None:
        pytest.skip("rocprofv3 not on PATH")
    script = tmp_path / "plans.py"
    script.write_text(_UNDER_PROFILER % {"root": ROOT})
    env = dict(os.environ, TMPDIR=str(tmp_path), MIFFT_JIT_VERBOSE="1")
    r = subprocess.run([prof, "--kernel-trace", "-d", str(tmp_path / "prof"), "--", sys.executable, str(script)],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout + r.stderr
    assert "KERNELS (8, 343) ['rows343_7x7x7_jit']" in out, out[-3000:]
    assert "_fs1_jit" in out and "runtime specialisation of" not in out, out[-3000:]
