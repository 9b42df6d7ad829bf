"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, the reference's
golden vectors, and size-independent properties at the BASELINE sizes.

Tolerances:
  * golden vectors: the reference's own atol=1e-2 / rtol=1e-5 (fft/tests.mojo:40-41);
  * vs the oracle (fp32): per-transform ||y - y_ref||_2 / ||y_ref||_2 <= 1e-5
    (BASELINE.json north_star); fp64: <= 1e-12.
"""
import numpy as np
import pytest
import torch

import hackathon_fft_amd as mf
from conftest import (REF_ATOL, check_hermitian_plan, REF_RTOL, REL_L2_TOL_F32, REL_L2_TOL_F64, from_complex, load_matrix, rel_l2,
                      to_complex)
from oracle import mifft_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TDT = {np.float32: torch.float32, np.float64: torch.float64, np.uint8: torch.uint8, np.int32: torch.int32}


def gpu_fft(x_np, *, bases=None, inverse=False, out_dtype=np.float64, faithful=False, first=0, count=None):
    """Run one transform through plan_fft / fft with NaN-prefilled output (fft/tests.mojo:219-222)."""
    x = torch.from_numpy(np.ascontiguousarray(x_np)).to(DEV)
    x_before = x.clone()
    out_shape = tuple(x.shape[:-1]) + (2,)
    out = torch.full(out_shape, float("nan"), dtype=TDT[out_dtype], device=DEV)
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(x.dtype, TDT[out_dtype], tuple(x.shape), out_shape, bases=bases, inverse=inverse,
                           faithful_stages=faithful, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan, first=first, count=count)
        ctx.synchronize()
    assert torch.equal(x, x_before), "x must never be written"
    return out.cpu().numpy(), plan


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("n,bases", load_matrix())
def test_reference_matrix_forward_real_fp64(golden_1d, n, bases, faithful):
    pairs = golden_1d["values"][str(n)]
    x = np.array([p["x"] for p in pairs], dtype=np.float64).reshape(len(pairs), n, 1)
    out, plan = gpu_fft(x, bases=[list(bases)], faithful=faithful)
    assert not np.isnan(out).any(), "every output element must be written"
    np.testing.assert_allclose(out, np.array([p["X"] for p in pairs]), atol=REF_ATOL, rtol=REF_RTOL)
    assert rel_l2(out, O.fftn(x, bases=[list(bases)])) < REL_L2_TOL_F64
    assert plan.stages(0) == O.ordered_bases(n, bases)


@pytest.mark.parametrize("n,bases", load_matrix())
def test_reference_matrix_inverse_complex_fp64(golden_1d, n, bases):
    pairs = golden_1d["values"][str(n)]
    spectrum = np.array([p["X"] for p in pairs], dtype=np.float64)
    series = np.array([p["x"] for p in pairs], dtype=np.float64)
    out, _ = gpu_fft(spectrum, bases=[list(bases)], inverse=True, faithful=True)
    np.testing.assert_allclose(out[..., 0], series, atol=REF_ATOL, rtol=REF_RTOL)
    np.testing.assert_allclose(out[..., 1], 0, atol=REF_ATOL, rtol=REF_RTOL)
    assert rel_l2(out, O.fftn(spectrum, inverse=True, bases=[list(bases)])) < 1e-11


@pytest.mark.parametrize("n,bases", load_matrix()[::3])
def test_reference_matrix_fp32(golden_1d, n, bases):
    pairs = golden_1d["values"][str(n)]
    x = np.array([p["x"] for p in pairs], dtype=np.float32).reshape(len(pairs), n, 1)
    for faithful in (True, False):
        out, _ = gpu_fft(x, bases=[list(bases)], out_dtype=np.float32, faithful=faithful)
        np.testing.assert_allclose(out, np.array([p["X"] for p in pairs]), atol=REF_ATOL, rtol=REF_RTOL)
        assert rel_l2(out, O.fftn(x, bases=[list(bases)])) < REL_L2_TOL_F32


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("which", ["2d", "3d"])
def test_nd_uint8_to_f64(golden_2d, golden_3d, which, faithful):
    g = golden_2d if which == "2d" else golden_3d
    x = np.array(g["x"], dtype=np.uint8)[None, ..., None]
    out, plan = gpu_fft(x, faithful=faithful)
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, np.array(g["X_flat"]).reshape(out.shape), atol=REF_ATOL, rtol=REF_RTOL)
    assert rel_l2(out, O.fftn(x, out_dtype=np.float64)) < REL_L2_TOL_F64
    assert plan.num_launches == len(g["shape"])  # no transpose launches (reference: d + 2(d-1))
    back, _ = gpu_fft(out, inverse=True, faithful=faithful)
    np.testing.assert_allclose(back[..., 0], x[..., 0], atol=1e-9)


# (shape, user bases) -- BASELINE.json configs at sizes the oracle finishes in seconds
BASELINE_CASES = [
    ((64, 128), None),
    ((37, 1024), [[2]]),
    ((301, 93), [[31, 3]]),
    ((2, 640, 480), None),
    ((2, 128, 128, 128), None),
    ((3, 64, 64, 64), None),     # fused 64x64 LDS plane + in-place column tiles
    ((70, 64, 64), None),        # 2-D through the fused plane kernel alone
    ((7, 128, 128), None),       # fused 128x128 plane (one workgroup per CU, next plane prefetched)
    ((300, 128, 128), None),     # more planes than workgroups: the persistent loop + prefetch path
    ((2, 256, 256), None),
    ((3, 512), None),
    ((3, 2048), None),
    ((5, 256), None),
    ((9, 64), None),
    ((5, 16, 12, 10), None),
    ((2, 6, 8, 10, 12), None),   # 4 transformed dims
    ((1, 3, 4, 64, 64), None),
    ((130, 480), None),
    ((90, 640), None),
    ((7, 4096), None),
    ((1, 2), None),
    ((3, 97), None),
]


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("shape,bases", BASELINE_CASES)
def test_c2c_fp32_vs_oracle(shape, bases, faithful):
    rng = np.random.default_rng(1234)
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = gpu_fft(x, bases=bases, out_dtype=np.float32, faithful=faithful)
    assert not np.isnan(out).any()
    ref = O.fftn(x, bases=bases)
    err = rel_l2(out, ref)
    assert err < REL_L2_TOL_F32, (shape, [plan.kernel_name(d) for d in range(len(shape) - 1)], err)
    truth = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    back, _ = gpu_fft(out, bases=bases, inverse=True, out_dtype=np.float32, faithful=faithful)
    assert rel_l2(back, x) < REL_L2_TOL_F32


@pytest.mark.parametrize("shape", [(9, 1024), (11, 93), (2, 40, 30), (1, 8, 6, 10), (5, 128), (3, 64), (3, 256),
                                   (3, 512), (1, 640, 480), (1, 128, 128, 128), (2, 64, 64, 64), (1, 256, 256)])
def test_c2c_fp64_vs_oracle(shape):
    rng = np.random.default_rng(99)
    x = rng.standard_normal(shape + (2,))
    ref = O.fftn(x)
    for faithful in (True, False):
        out, plan = gpu_fft(x, faithful=faithful)
        assert rel_l2(out, ref) < REL_L2_TOL_F64, [plan.kernel_name(d) for d in range(len(shape) - 1)]
        back, _ = gpu_fft(out, inverse=True, faithful=faithful)
        assert rel_l2(back, x) < REL_L2_TOL_F64


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(8, 1024), (6, 93), (9, 128), (2, 640, 480), (2, 64, 64, 64), (1, 128, 128, 128)])
def test_real_input_full_spectrum_on_the_fused_kernels(shape, dtype):
    """C_in = 1 (fft/fft/_fft.mojo:254-255): the fused kernels promote a real row in their HBM load."""
    rng = np.random.default_rng(17)
    x = rng.standard_normal(shape + (1,)).astype(dtype)
    out, plan = gpu_fft(x, out_dtype=dtype)
    assert plan.kernel_name(len(shape) - 2) != "generic"
    tol = REL_L2_TOL_F32 if dtype == np.float32 else REL_L2_TOL_F64
    assert rel_l2(out, O.fftn(x, out_dtype=dtype)) < tol
    truth = np.fft.fftn(x[..., 0].astype(np.float64), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < tol


@pytest.mark.parametrize("in_dtype", [np.float32, np.float64, np.uint8, np.int32])
@pytest.mark.parametrize("comps", [1, 2])
def test_input_dtypes_and_real_input(in_dtype, comps):
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 200, size=(4, 12, 20, comps))).astype(in_dtype)
    out, _ = gpu_fft(x, out_dtype=np.float32)
    ref = O.fftn(x, out_dtype=np.float32)
    assert rel_l2(out, ref) < REL_L2_TOL_F32


def _narrow_input(kind, shape, seed):
    """(torch tensor on the CPU, numpy array for the oracle, oracle in_dtype or None, float64 values)"""
    rng = np.random.default_rng(seed)
    if kind == "bfloat16":
        t = torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * 8).to(torch.bfloat16)
        return t, t.view(torch.int16).numpy().view(np.uint16), O.BF16, t.to(torch.float64).numpy()
    if kind == "float16":
        a = (rng.standard_normal(shape) * 8).astype(np.float16)
    elif kind == "int8":
        a = rng.integers(-128, 128, size=shape).astype(np.int8)
    elif kind == "int16":
        a = rng.integers(-32768, 32768, size=shape).astype(np.int16)
    else:
        a = rng.integers(0, 65536, size=shape).astype(np.uint16)
    return torch.from_numpy(a), a, None, a.astype(np.float64)


@pytest.mark.parametrize("kind", ["int8", "int16", "uint16", "float16", "bfloat16"])
@pytest.mark.parametrize("comps", [1, 2])
@pytest.mark.parametrize("shape,faithful", [((5, 96), False), ((3, 1024), False), ((2, 24, 20), False), ((2, 6, 4, 8), False),
                                            ((3, 60), True)])
def test_every_input_element_type_the_reference_casts(kind, comps, shape, faithful):
    """The reference widens ANY element type in its first-stage load (`x.load(...).cast[out_dtype]()`,
    fft/fft/_fft.mojo:243-257; its own tests feed uint8).  int8 / int16 / uint16 / half / bfloat16, real and complex,
    through the runtime-specialised first pass (and the literal stages with faithful=True), against the oracle with the
    same casts and against fp64 pocketfft on the exactly widened values."""
    if kind == "uint16" and not hasattr(torch, "uint16"):
        pytest.skip("this torch has no uint16")
    t, a, ora_dtype, vals = _narrow_input(kind, shape + (comps,), seed=len(kind) + comps + sum(shape))
    for odt, tol in ((torch.float32, REL_L2_TOL_F32), (torch.float64, REL_L2_TOL_F64)):
        x = t.to(DEV)
        out = torch.full(shape + (2,), float("nan"), dtype=odt, device=DEV)
        with mf.DeviceContext(0) as ctx:
            plan = mf.plan_fft(x.dtype, odt, tuple(x.shape), tuple(out.shape), faithful_stages=faithful, ctx=ctx)
            mf.fft(out, x, ctx, plan=plan)
            ctx.synchronize()
        got = out.cpu().numpy()
        assert not np.isnan(got).any()
        ndt = np.float32 if odt == torch.float32 else np.float64
        assert rel_l2(got, O.fftn(a, out_dtype=ndt, in_dtype=ora_dtype)) < tol, plan.kernel_name(len(shape) - 2)
        z = vals[..., 0] + (1j * vals[..., 1] if comps == 2 else 0)
        truth = np.fft.fftn(z, axes=tuple(range(1, len(shape))))
        assert rel_l2(got, from_complex(truth, np.float64)) < (REL_L2_TOL_F32 if odt == torch.float32 else 1e-11)
        if not faithful:
            assert "_jit" in plan.kernel_name(len(shape) - 2), plan.kernel_name(len(shape) - 2)


@pytest.mark.parametrize("shape,dtype", [((25, 640, 480), np.float32), ((50, 64, 64, 64), np.float32), ((10, 128, 128, 128), np.float32),
                                         ((1, 256, 256, 256), np.float32), ((30, 360, 280), np.float64),
                                         ((10, 1920, 1080), np.float32),      # 8-column tiles: no carried column
                                         ((2, 64, 64, 64, 64), np.float32)])  # three trailing dimensions
@pytest.mark.parametrize("inverse", [False, True])
def test_last_pass_of_real_input_plans_uses_the_hermitian_symmetry(shape, dtype, inverse):
    """Real input, 2-D .. 4-D: the last (strided, in-place) pass transforms only the columns up to their mirror and stores
    every result twice, at (k, c) and conjugated at (-k, -c) (TileCfg::HERM; the reference computes all of it,
    fft/fft/_ndim_fft_gpu.mojo:634-642).  Shapes large enough for the plan-time policy (herm_pays, mifft_internal.h) to take
    the twin; tests/test_gpu_lab.py forces it on small, ragged and odd shapes.  Checks: conftest.check_hermitian_plan."""
    check_hermitian_plan(gpu_fft, shape, dtype, inverse)


def test_batch_range_and_untouched_rows():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((10, 1024, 2)).astype(np.float32)
    full, _ = gpu_fft(x, out_dtype=np.float32)
    part, _ = gpu_fft(x, out_dtype=np.float32, first=3, count=4)
    assert np.isnan(part[:3]).all() and np.isnan(part[7:]).all()
    assert np.array_equal(part[3:7], full[3:7])  # a slab equals the same rows of the whole batch, bit for bit
    empty, _ = gpu_fft(x, out_dtype=np.float32, first=0, count=0)
    assert np.isnan(empty).all()


@pytest.mark.parametrize("shape", [(25, 640, 480), (12, 128, 128, 128), (3, 256, 256, 256)])
def test_batch_range_of_a_real_input_plan_with_hermitian_passes(shape):
    """mifft_exec_batch(first, count) on plans whose passes skip / mirror halves of the tensor (`_hs`, `_h`): the rows inside
    the range equal the whole-batch result bit for bit, the rows outside are not touched -- the mirrored stores of the last
    pass stay inside their own image."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (1,)).astype(np.float32)
    full, plan = gpu_fft(x, out_dtype=np.float32)
    names = [plan.kernel_name(d) for d in range(len(shape) - 1)]
    assert names[0].endswith(("_h", "_h_jit")) and any("_hs" in n for n in names[1:]), names
    first, count = 1, max(1, shape[0] // 2)
    part, _ = gpu_fft(x, out_dtype=np.float32, first=first, count=count)
    assert np.isnan(part[:first]).all() and np.isnan(part[first + count:]).all()
    assert np.array_equal(part[first:first + count], full[first:first + count])


@pytest.mark.parametrize("shape", [(10, 1024), (37, 93), (21, 128), (19, 480), (3, 40, 64)])
def test_results_do_not_depend_on_the_tile_slot(shape):
    """The same transform placed at every position of the batch gives bit-identical output (FMA placement is
    pinned in fft_radix.h), so splitting the batch over tiles, slabs or GPUs cannot change a single bit."""
    rng = np.random.default_rng(23)
    one = rng.standard_normal((1,) + shape[1:] + (2,)).astype(np.float32)
    x = np.repeat(one, shape[0], axis=0)
    out, plan = gpu_fft(x, out_dtype=np.float32)
    assert plan.kernel_name(len(shape) - 2) != "generic"
    for i in range(1, shape[0]):
        assert np.array_equal(out[i], out[0]), i
    again, _ = gpu_fft(x, out_dtype=np.float32)
    assert np.array_equal(again, out)


def test_errors_from_the_device_side_of_the_boundary():
    x = torch.zeros((2, 8, 2), device=DEV)
    plan = mf.plan_fft(torch.float32, torch.float32, (2, 8, 2), (2, 8, 2))
    with pytest.raises(mf.MifftError) as e:
        mf.fft(x, x, plan=plan)           # aliasing: the reference is out-of-place
    assert e.value.status == -13
    with pytest.raises(mf.MifftError):
        mf.fft(torch.zeros((2, 8, 2)), x, plan=plan)   # host tensor
    with pytest.raises(mf.MifftError):
        mf.fft(torch.zeros((2, 9, 2), device=DEV), x, plan=plan)
    with pytest.raises(mf.MifftError) as e:
        mf.plan_fft(torch.float32, torch.float32, (1, 1 << 25, 2), (1, 1 << 25, 2), bases=[[2]])
    assert e.value.status == -9            # beyond four-step reach (4096 x 4096 points)


@pytest.mark.parametrize("shape", [(7, 1024, 3), (5, 640, 8), (3, 128, 5, 2), (4, 343, 7), (9, 64, 2), (6, 97, 3)])
def test_strided_dimension_with_fewer_columns_than_a_tile(shape):
    """A transformed dimension in front of a very short one (3 channels, 2 components): the 16-column tile is ragged from
    the start, and still the fused kernels run (not the literal stages)."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = gpu_fft(x, out_dtype=np.float32)
    assert plan.kernel_name(0) != "generic", [plan.kernel_name(d) for d in range(len(shape) - 1)]
    truth = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    back, _ = gpu_fft(out, inverse=True, out_dtype=np.float32)
    assert rel_l2(back, x) < REL_L2_TOL_F32


def test_convenience_wrappers_accept_any_length():
    """plan_fft rejects lengths its (reference) default radices cannot factor; fftn / ifftn / rfftn retry with the full
    prime factorisation, numpy-style."""
    rng = np.random.default_rng(4)
    for shape in [(3, 634), (2, 131, 6), (2, 1009)]:          # 2 * 317, prime 131, prime 1009
        xc = torch.from_numpy(rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).to(DEV)
        with pytest.raises(mf.MifftError):
            mf.plan_fft(torch.float64, torch.float64, tuple(shape) + (2,), tuple(shape) + (2,))
        y = mf.fftn(xc)
        np.testing.assert_allclose(y.cpu().numpy(), np.fft.fftn(xc.cpu().numpy(), axes=tuple(range(1, len(shape)))),
                                   atol=1e-9)
        np.testing.assert_allclose(mf.ifftn(y).cpu().numpy(), xc.cpu().numpy(), atol=1e-11)


def test_convenience_wrappers():
    rng = np.random.default_rng(2)
    xc = torch.from_numpy(rng.standard_normal((3, 16, 12)) + 1j * rng.standard_normal((3, 16, 12))).to(DEV)
    y = mf.fftn(xc, radices=[[4, 2], [3, 2]])
    np.testing.assert_allclose(y.cpu().numpy(), np.fft.fftn(xc.cpu().numpy(), axes=(1, 2)), atol=1e-10)
    np.testing.assert_allclose(mf.ifftn(y).cpu().numpy(), xc.cpu().numpy(), atol=1e-12)
    xr = torch.from_numpy(rng.standard_normal((2, 30))).to(DEV)
    full = mf.rfftn(xr)
    np.testing.assert_allclose(to_complex(full.cpu().numpy()), np.fft.fft(xr.cpu().numpy(), axis=1), atol=1e-12)
    # wrappers are asynchronous on the current stream and reuse cached plans
    from hackathon_fft_amd import api
    before = len(api._PLAN_CACHE)
    ys = [mf.fftn(xc, radices=[[4, 2], [3, 2]]) for _ in range(20)]
    assert len(api._PLAN_CACHE) == before
    assert all(torch.equal(ys[0], t) for t in ys[1:])
    for n in range(2, 2 + api._PLAN_CACHE_SIZE + 5):   # eviction path
        mf.fftn(torch.zeros((1, n), dtype=torch.complex64, device=DEV))
    assert len(api._PLAN_CACHE) == api._PLAN_CACHE_SIZE
    mf.clear_plan_cache()
    assert len(api._PLAN_CACHE) == 0
    # one cached plan PER STREAM: a plan's scratch / counters allow one exec in flight, so two streams transforming the
    # same shape must not share one
    big = torch.from_numpy(rng.standard_normal((1, 5120, 8)) + 1j * rng.standard_normal((1, 5120, 8))).to(DEV)
    ref = np.fft.fftn(big.cpu().numpy(), axes=(1, 2))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for st in (s1, s2, s1, s2):
        with torch.cuda.stream(st):
            outs.append(mf.fftn(big))           # the long strided dimension runs through the plan-owned scratch
    torch.cuda.synchronize()
    assert len(api._PLAN_CACHE) == 2
    for o in outs:
        np.testing.assert_allclose(o.cpu().numpy(), ref, atol=1e-8)
    mf.clear_plan_cache()


# ---- rank 4 and 5 (the reference takes any rank > 2, fft/fft/fft.mojo:20-46; its bench list holds the 4-D 64^4 and the
#      5-D 25 x 160 x 160 x 48, fft/bench.mojo:120-121) ----

@pytest.mark.parametrize("shape,comps,dtype", [((2, 4, 6, 8, 10), 2, np.float32), ((1, 3, 4, 5, 6, 7), 2, np.float64),
                                               ((2, 5, 16, 16, 12), 1, np.float32), ((1, 2, 3, 4, 5, 6, 7), 2, np.float32),
                                               ((3, 7, 9, 11, 13), 2, np.float64), ((1, 6, 10, 64, 64), 1, np.float64)])
def test_rank_4_to_6_vs_oracle(shape, comps, dtype):
    rng = np.random.default_rng(len(shape) * 1000 + shape[-1])
    x = rng.standard_normal(shape + (comps,)).astype(dtype)
    out, plan = gpu_fft(x, out_dtype=dtype)
    tol = REL_L2_TOL_F32 if dtype == np.float32 else REL_L2_TOL_F64
    assert not np.isnan(out).any()
    assert rel_l2(out, O.fftn(x, out_dtype=dtype)) < tol
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < tol
    back, _ = gpu_fft(out, inverse=True, out_dtype=dtype)
    want = x if comps == 2 else np.concatenate([x, np.zeros_like(x)], axis=-1)
    assert rel_l2(back, want) < tol


@pytest.mark.parametrize("shape,comps", [((1, 64, 64, 64, 64), 2), ((1, 25, 160, 160, 48), 1), ((1, 25, 160, 160, 48), 2)])
def test_reference_bench_shapes_of_rank_4_and_5(shape, comps):
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(shape + (comps,), generator=g, device=DEV, dtype=torch.float32)
    out = torch.full(shape + (2,), float("nan"), device=DEV)
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    assert not torch.isnan(out).any()
    xh = x.cpu().numpy().astype(np.float64)
    xc = xh[..., 0] + (1j * xh[..., 1] if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    assert rel_l2(out.cpu().numpy(), from_complex(truth, np.float64)) < REL_L2_TOL_F32
    inv = mf.plan_fft(torch.float32, torch.float32, out.shape, out.shape, inverse=True, ctx=ctx)
    back = torch.full_like(out, float("nan"))
    mf.fft(back, out, ctx, plan=inv)
    ctx.synchronize()
    want = x if comps == 2 else torch.cat([x, torch.zeros_like(x)], dim=-1)
    assert ((back - want).norm() / want.norm()).item() < 1e-5


# ---- full BASELINE sizes: properties that need no CPU reference ----------------------------

FULL_SIZES = [
    ((100000, 1024), [[2]]),
    ((500000, 93), [[31, 3]]),
    ((500000, 128), None),
    ((100, 640, 480), None),
    ((10, 128, 128, 128), None),
]


@pytest.mark.parametrize("shape,bases", FULL_SIZES)
def test_full_size_properties(shape, bases):
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn(shape + (2,), generator=g, device=DEV, dtype=torch.float32)
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    fwd = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=bases, ctx=ctx)
    inv = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=bases, inverse=True, ctx=ctx)
    mf.fft(out, x, ctx, plan=fwd)
    ctx.synchronize()
    assert not torch.isnan(out).any()
    n_per = int(np.prod(shape[1:]))
    b = shape[0]
    # Parseval per transform: sum |X|^2 = N sum |x|^2
    ex = (x.double() ** 2).reshape(b, -1).sum(1)
    eX = (out.double() ** 2).reshape(b, -1).sum(1)
    assert ((eX / (n_per * ex) - 1).abs().max().item()) < 1e-5
    # DC bin = sum of the inputs
    dc = out.reshape(b, n_per, 2)[:, 0, :].double()
    s = x.double().reshape(b, n_per, 2).sum(1)
    assert ((dc - s).norm(dim=1) / (ex.sqrt() * np.sqrt(n_per))).max().item() < 1e-5
    # encode -> decode round trip
    back = torch.full_like(x, float("nan"))
    mf.fft(back, out, ctx, plan=inv)
    ctx.synchronize()
    num = (back.double() - x.double()).reshape(b, -1).norm(dim=1)
    assert (num / ex.sqrt()).max().item() < 1e-5
    # spot-check against the oracle: one transform per 4096 (at least a dozen spread over the batch), plus the first
    # and last transforms of the ragged last tile for every tile size the kernels use (4, 8, 16, 32, 64 transforms)
    step = max(1, min(4096, b // 12))
    idx = set(range(0, b, step)) | {0, 1, b - 2, b - 1}
    for tile in (4, 8, 16, 32, 64):
        idx |= {((b - 1) // tile) * tile, max(0, ((b - 1) // tile) * tile - 1)}
    idx = sorted(i for i in idx if 0 <= i < b)
    sel = torch.tensor(idx, device=DEV)
    xs = x.index_select(0, sel).cpu().numpy()
    small = O.plan_fft(np.float32, np.float32, xs.shape, xs.shape, bases=bases)
    ref = np.empty_like(xs)
    O.fft(ref, xs, plan=small)
    got = out.index_select(0, sel).cpu().numpy()
    for k in range(len(idx)):
        assert rel_l2(got[k:k + 1], ref[k:k + 1]) < REL_L2_TOL_F32, (shape, idx[k])
    # linearity on a slab: F(a*x + y) = a*F(x) + F(y)
    k = min(b, 64)
    y = torch.randn((k,) + tuple(x.shape[1:]), generator=g, device=DEV, dtype=torch.float32)
    lin_plan = mf.plan_fft(torch.float32, torch.float32, y.shape, y.shape, bases=bases, ctx=ctx)
    fy, fz = torch.empty_like(y), torch.empty_like(y)
    z = (0.5 * x[:k] + y).contiguous()
    mf.fft(fy, y, ctx, plan=lin_plan)
    mf.fft(fz, z, ctx, plan=lin_plan)
    ctx.synchronize()
    lhs, rhs = fz.double(), 0.5 * out[:k].double() + fy.double()
    assert ((lhs - rhs).reshape(k, -1).norm(dim=1) / rhs.reshape(k, -1).norm(dim=1)).max().item() < 1e-5


# the reference's own benchmark is the real-input one (bench_gpu_radix_n_rfft, fft/bench.mojo:57-97,108-124): its shapes
REAL_FULL_SIZES = [(250000, 93), (250000, 128), (100000, 1024), (100, 640, 480), (100, 64, 64, 64), (10, 128, 128, 128)]


@pytest.mark.parametrize("shape", REAL_FULL_SIZES)
def test_full_size_real_input(shape):
    """C_in = 1 at the reference's benchmark sizes (streaming `_r_nt` twins, real-input planes): equal to the complex
    plan on (x, 0) within rounding, Hermitian-symmetric, and oracle spot-checks incl. the ragged last tile."""
    g = torch.Generator(device=DEV).manual_seed(4321)
    x = torch.randn(shape + (1,), generator=g, device=DEV, dtype=torch.float32)
    out = torch.full(shape + (2,), float("nan"), device=DEV)
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    assert not torch.isnan(out).any()
    assert "generic" not in [plan.kernel_name(d) for d in range(len(shape) - 1)]
    xc = torch.cat([x, torch.zeros_like(x)], dim=-1).contiguous()
    outc = torch.empty_like(out)
    cplan = mf.plan_fft(torch.float32, torch.float32, xc.shape, out.shape, ctx=ctx)
    mf.fft(outc, xc, ctx, plan=cplan)
    ctx.synchronize()
    b = shape[0]
    num = (out.double() - outc.double()).reshape(b, -1).norm(dim=1)
    den = outc.double().reshape(b, -1).norm(dim=1)
    assert (num / den).max().item() < 2e-6
    # Hermitian symmetry X[-k] = conj(X[k]) on a slab of the batch (index reversal along every transformed dimension)
    k = min(b, 16)
    z = torch.view_as_complex(out[:k].contiguous())
    dims = tuple(range(1, z.dim()))
    zr = torch.roll(torch.flip(z, dims), shifts=[1] * len(dims), dims=dims)
    assert ((zr.conj() - z).abs().reshape(k, -1).norm(dim=1) / z.abs().reshape(k, -1).norm(dim=1)).max().item() < 2e-6
    step = max(1, b // 6)
    idx = sorted({0, b - 1} | set(range(0, b, step)) | {((b - 1) // 64) * 64})
    sel = torch.tensor(idx, device=DEV)
    xs = x.index_select(0, sel).cpu().numpy()
    ref = O.fftn(xs)
    got = out.index_select(0, sel).cpu().numpy()
    for i in range(len(idx)):
        assert rel_l2(got[i:i + 1], ref[i:i + 1]) < REL_L2_TOL_F32, (shape, idx[i])


@pytest.mark.parametrize("shape,comps", [((3, 16384), 2), ((1, 8192), 2), ((261, 16384), 2), ((515, 8192), 2), ((2, 3, 16384), 2),
                                         ((5, 2, 8192), 2), ((7, 16384), 1), ((130, 16384), 1)])
def test_long_rows_as_a_four_step_inside_lds(shape, comps):
    """plane_kernel_wp<FS>: 16384 = 128 x 128 and 8192 = 64 x 128 points per row in one launch"""
    rng = np.random.default_rng(shape[0] + shape[-1])
    x = rng.standard_normal(shape + (comps,)).astype(np.float32)
    out, plan = gpu_fft(x, out_dtype=np.float32)
    assert "_fs" in plan.kernel_name(len(shape) - 2) and "_wp" in plan.kernel_name(len(shape) - 2), plan.kernel_name(len(shape) - 2)
    assert not np.isnan(out).any()
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    sel = [0, shape[0] - 1]
    assert rel_l2(out[sel], O.fftn(x[sel], out_dtype=np.float32)) < REL_L2_TOL_F32   # the oracle on the first and last transform
    back, _ = gpu_fft(out, inverse=True, out_dtype=np.float32)
    assert rel_l2(back, x if comps == 2 else np.concatenate([x, np.zeros_like(x)], axis=-1)) < REL_L2_TOL_F32
    # ragged ranges of the batch through the same plan
    if shape[0] > 4:
        part, _ = gpu_fft(x, out_dtype=np.float32, first=2, count=shape[0] - 3)
        assert np.array_equal(part[2:shape[0] - 1], out[2:shape[0] - 1]) and np.isnan(part[:2]).all() and np.isnan(part[-1:]).all()


@pytest.mark.parametrize("shape", [(2, 8192), (133, 8192), (3, 2, 8192)])
def test_long_fp64_rows_as_a_four_step_inside_lds(shape):
    rng = np.random.default_rng(shape[0])
    x = rng.standard_normal(shape + (2,))
    out, plan = gpu_fft(x, out_dtype=np.float64)
    assert "_fs" in plan.kernel_name(len(shape) - 2), plan.kernel_name(len(shape) - 2)
    truth = np.fft.fftn(x[..., 0] + 1j * x[..., 1], axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F64
    assert rel_l2(out[:1], O.fftn(x[:1], out_dtype=np.float64)) < REL_L2_TOL_F64
    back, _ = gpu_fft(out, inverse=True, out_dtype=np.float64)
    assert rel_l2(back, x) < REL_L2_TOL_F64


# batches in the 0.25-0.55 GB window take the non-temporal-store twins: table, generated table,
# runtime-specialised lengths, fp64, and planes that are a 2-D plan's only pass
WINDOW_CASES = [((250000, 93), torch.float32), ((250000, 128), torch.float32), ((30000, 1024), torch.float32),
                ((25000, 1000), torch.float32), ((40000, 625), torch.float32), ((100000, 128), torch.float64),
                ((12000, 1000), torch.float64), ((6400, 64, 64), torch.float32), ((1600, 128, 128), torch.float32),
                ((2600, 100, 100), torch.float32)]


def test_store_policy_skips_short_unaligned_runs():
    """17 x 17: the last pass stores 136-byte runs directly; non-temporal partial-line writes cost 20-30 % there"""
    ctx = mf.DeviceContext(0)
    full = (80000, 289, 2)
    plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
    assert "_nts" not in plan.kernel_name(0), plan.kernel_name(0)


@pytest.mark.parametrize("shape,dtype", WINDOW_CASES)
def test_store_policy_window_twins(shape, dtype):
    g = torch.Generator(device=DEV).manual_seed(31)
    x = torch.randn(shape + (2,), generator=g, device=DEV, dtype=dtype)
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(dtype, dtype, x.shape, x.shape, ctx=ctx)
    assert "_nts" in plan.kernel_name(len(shape) - 2), [plan.kernel_name(d) for d in range(len(shape) - 1)]
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    assert not torch.isnan(out).any()
    b = shape[0]
    idx = sorted({0, 1, b // 3, b // 2, b - 2, b - 1, ((b - 1) // 64) * 64})
    sel = torch.tensor(idx, device=DEV)
    xs = x.index_select(0, sel).cpu().numpy()
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    ref = O.fftn(xs, out_dtype=np_dt)
    got = out.index_select(0, sel).cpu().numpy()
    tol = REL_L2_TOL_F32 if dtype == torch.float32 else REL_L2_TOL_F64
    for i in range(len(idx)):
        assert rel_l2(got[i:i + 1], ref[i:i + 1]) < tol, (shape, idx[i])
    inv = mf.plan_fft(dtype, dtype, x.shape, x.shape, inverse=True, ctx=ctx)
    back = torch.full_like(x, float("nan"))
    mf.fft(back, out, ctx, plan=inv)
    ctx.synchronize()
    err = ((back.double() - x.double()).reshape(b, -1).norm(dim=1) / x.double().reshape(b, -1).norm(dim=1)).max().item()
    assert err < (1e-5 if dtype == torch.float32 else 1e-13)


@pytest.mark.parametrize("shape,bases", [((20000, 1024), [[2]]), ((150000, 128), None), ((210000, 93), [[31, 3]])])
def test_streaming_size_fp64(shape, bases):
    """fp64 tensors beyond the Infinity Cache take the non-temporal twins of the fp64 row kernels."""
    g = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randn(shape + (2,), generator=g, device=DEV, dtype=torch.float64)
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    fwd = mf.plan_fft(torch.float64, torch.float64, x.shape, x.shape, bases=bases, ctx=ctx)
    inv = mf.plan_fft(torch.float64, torch.float64, x.shape, x.shape, bases=bases, inverse=True, ctx=ctx)
    assert "_nt" in fwd.kernel_name(0), fwd.kernel_name(0)
    mf.fft(out, x, ctx, plan=fwd)
    ctx.synchronize()
    assert not torch.isnan(out).any()
    b, n = shape
    ex = (x ** 2).reshape(b, -1).sum(1)
    eX = (out ** 2).reshape(b, -1).sum(1)
    assert ((eX / (n * ex) - 1).abs().max().item()) < 1e-12
    back = torch.full_like(x, float("nan"))
    mf.fft(back, out, ctx, plan=inv)
    ctx.synchronize()
    assert ((back - x).reshape(b, -1).norm(dim=1) / ex.sqrt()).max().item() < 1e-13
    idx = sorted({0, 1, b // 3, b // 2, b - 2, b - 1, ((b - 1) // 64) * 64})
    sel = torch.tensor(idx, device=DEV)
    xs = x.index_select(0, sel).cpu().numpy()
    ref = O.fftn(xs, bases=bases, out_dtype=np.float64)
    got = out.index_select(0, sel).cpu().numpy()
    for i in range(len(idx)):
        assert rel_l2(got[i:i + 1], ref[i:i + 1]) < REL_L2_TOL_F64, (shape, idx[i])


@pytest.mark.parametrize("n,batch,dtype", [(32768, 3, np.float32), (65536, 2, np.float32), (1 << 20, 1, np.float32),
                                           (100000, 3, np.float32), (98304, 2, np.float32), (20480, 5, np.float32),
                                           (50000, 2, np.float64), (1 << 17, 1, np.float64), (1 << 22, 1, np.float32),
                                           (1 << 24, 1, np.float32), (40960, 3, np.float32), (1 << 23, 2, np.float32),
                                           (16384, 260, np.float32)])   # big batch of LDS-sized rows: two passes preferred
def test_four_step_large_dimension(n, batch, dtype):
    """Dimensions beyond one workgroup's LDS row: column FFTs with a transposed + twiddled
    store, then column FFTs in place (two launches, no scratch) when the first factor has such a kernel; otherwise
    column FFTs, transpose + twiddle, column FFTs.  Checked against fp64 pocketfft (the oracle needs minutes at
    these lengths)."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((batch, n, 2)).astype(dtype)
    out, plan = gpu_fft(x, out_dtype=dtype)
    assert not np.isnan(out).any()
    if "_fs" in plan.kernel_name(0):   # 16384 points: the four-step runs inside one LDS plane, one launch
        assert plan.num_launches == 1 and n == 16384
    else:
        assert plan.num_launches == (2 if "_ts" in plan.kernel_name(0) else 3)
        if dtype == np.float32 and n & (n - 1) == 0:
            assert plan.num_launches == 2
    truth = np.fft.fft(to_complex(x), axis=1)
    tol = REL_L2_TOL_F32 if dtype == np.float32 else 1e-11
    assert rel_l2(out, from_complex(truth, np.float64)) < tol
    back, _ = gpu_fft(out, inverse=True, out_dtype=dtype)
    assert rel_l2(back, x) < tol
    if batch > 1:   # a slab of the batch through the same plan
        part, _ = gpu_fft(x, out_dtype=dtype, first=1, count=1)
        assert np.isnan(part[0]).all() and np.array_equal(part[1], out[1])


@pytest.mark.parametrize("n,in_dtype,comps,out_dtype", [(1 << 20, np.float32, 1, np.float32), (1 << 18, np.uint8, 2, np.float32),
                                                        (1 << 17, np.int32, 1, np.float32), (98304, np.float32, 1, np.float32),
                                                        (1 << 18, np.float32, 2, np.float64), (1 << 19, np.float64, 1, np.float64)])
def test_four_step_real_integer_and_mixed_input(n, in_dtype, comps, out_dtype):
    """The first pass of the two-launch four-step reads x with its own element type and component count (a
    runtime-specialised transposed-store kernel), so long transforms are not limited to complex input of the output
    dtype."""
    rng = np.random.default_rng(n + comps)
    if np.issubdtype(in_dtype, np.integer):
        x = rng.integers(0, 200, size=(2, n, comps)).astype(in_dtype)
    else:
        x = rng.standard_normal((2, n, comps)).astype(in_dtype)
    out, plan = gpu_fft(x, out_dtype=out_dtype)
    assert plan.num_launches == 2 and "_ts" in plan.kernel_name(0), plan.kernel_name(0)
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    truth = np.fft.fft(xc, axis=1)
    tol = REL_L2_TOL_F32 if out_dtype == np.float32 else 1e-11
    assert rel_l2(out, from_complex(truth, np.float64)) < tol


@pytest.mark.parametrize("shape,comps", [((2, 6, 32768), 2), ((1, 3, 5, 20480), 2), ((2, 4, 65536), 1), ((1, 64, 1 << 17), 2)])
def test_four_step_on_the_contiguous_dimension_of_an_nd_transform(shape, comps):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (comps,)).astype(np.float32)
    out, plan = gpu_fft(x, out_dtype=np.float32)
    last = len(shape) - 2
    assert "_ts" in plan.kernel_name(last), plan.kernel_name(last)
    assert plan.num_launches == len(shape)      # two for the long dimension, one per other dimension
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    truth = np.fft.fftn(xc, axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    if comps == 2:
        back, _ = gpu_fft(out, inverse=True, out_dtype=np.float32)
        assert rel_l2(back, x) < REL_L2_TOL_F32


LONG_STRIDED = [(1, 7680, 64), (2, 5120, 40), (1, 8192, 3, 5), (1, 6144, 100), (2, 5000, 24), (1, 9000, 10),
                (1, 4320, 7680), (3, 16384, 16), (2, 2, 4608, 33)]


@pytest.mark.parametrize("shape", LONG_STRIDED)
def test_long_strided_dimension_four_step(shape):
    """A strided dimension beyond the column-tile table (8K-video columns): two column passes through the plan scratch
    (N = N1 * N2: FS1 tiles store row k1 of (n2, columns) as row n2 * N1 + k1 times W^(k1 n2), then N2-point tiles)."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = gpu_fft(x, out_dtype=np.float32)
    long_dim = int(np.argmax(shape[1:]))
    if long_dim < len(shape) - 2:
        assert "_fs1" in plan.kernel_name(long_dim), plan.kernel_name(long_dim)
        assert plan.scratch_bytes == x.nbytes
    truth = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    back, _ = gpu_fft(out, inverse=True, out_dtype=np.float32)
    assert rel_l2(back, x) < REL_L2_TOL_F32


