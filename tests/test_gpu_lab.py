"""Tests of the LAB build of the library (hackathon_fft_amd/csrc/libmifft_lab.so: -DMIFFT_EXPERIMENTAL -DMIFFT_TESTING).

The product library carries neither the experimental kernels that stayed negative results (L2-resident image kernel,
wave-shuffle radix 3), nor the switches that force a fallback route, nor the allocation-failure injection; they are
parity-tested here against the lab build, loaded as a SECOND instance of the host package bound to the other shared
library (ctypes handles are process-local, so both libraries live side by side in this process)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REL_L2_TOL_F32, ROOT, check_hermitian_plan, from_complex, rel_l2, to_complex
from oracle import mifft_oracle as O

pytestmark = pytest.mark.gpu

LAB_LIB = os.path.join(ROOT, "hackathon_fft_amd", "csrc", "libmifft_lab.so")
_lab = None


def lab():
    """hackathon_fft_amd imported a second time under another name, with MIFFT_LIBRARY pointing at the lab build."""
    global _lab
    if _lab is None:
        if not os.path.exists(LAB_LIB):
            pytest.fail(f"{LAB_LIB} not built (make -C hackathon_fft_amd/csrc)")
        pkg = os.path.join(ROOT, "hackathon_fft_amd")
        spec = importlib.util.spec_from_file_location("hackathon_fft_amd_lab", os.path.join(pkg, "__init__.py"),
                                                      submodule_search_locations=[pkg])
        mod = importlib.util.module_from_spec(spec)
        old = os.environ.get("MIFFT_LIBRARY")
        os.environ["MIFFT_LIBRARY"] = LAB_LIB
        try:
            sys.modules["hackathon_fft_amd_lab"] = mod
            spec.loader.exec_module(mod)
            assert mod.LIB_PATH == LAB_LIB
            mod._lib.lib()
        finally:
            if old is None:
                del os.environ["MIFFT_LIBRARY"]
            else:
                os.environ["MIFFT_LIBRARY"] = old
        _lab = mod
    return _lab


def lab_fft(x_np, *, inverse=False, first=0, count=None):
    mf = lab()
    x = torch.from_numpy(np.ascontiguousarray(x_np)).to("cuda:0")
    out = torch.full(tuple(x.shape[:-1]) + (2,), float("nan"), dtype=x.dtype, device="cuda:0")
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(x.dtype, x.dtype, tuple(x.shape), tuple(out.shape), inverse=inverse, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan, first=first, count=count)
        ctx.synchronize()
    return out.cpu().numpy(), plan


def test_the_product_library_has_no_lab_switches(monkeypatch):
    """The same switches set for the PRODUCT library change nothing: its policy constants are compiled in."""
    import hackathon_fft_amd as mf
    monkeypatch.setenv("MIFFT_FOURSTEP_STRIDED", "0")
    monkeypatch.setenv("MIFFT_DPP", "1")
    monkeypatch.setenv("MIFFT_TEST_FAIL_SCRATCH_ALLOC", "1")
    plan = mf.plan_fft(torch.float32, torch.float32, (1, 7680, 64, 2), (1, 7680, 64, 2), ctx=mf.DeviceContext(0))
    assert "_fs1" in plan.kernel_name(0) and plan.scratch_bytes == 7680 * 64 * 8
    plan = mf.plan_fft(torch.float32, torch.float32, (16, 93, 2), (16, 93, 2), ctx=mf.DeviceContext(0))
    assert "dpp" not in plan.kernel_name(0)


@pytest.mark.parametrize("shape", [(20, 640, 480), (33, 256, 512)])
def test_l2_resident_image_kernel_opt_in(shape, monkeypatch):
    """MIFFT_JIT_IMAGE=1: every XCD transforms whole images -- rows x -> out, an XCD-local barrier (hardware XCC_ID,
    relaxed agent-scope atomics in the shared L2), columns in place from L2.  Parity only: on MI355X the path is slower
    than the two streaming passes (DESIGN_EXPERIMENTS.md)."""
    monkeypatch.setenv("MIFFT_JIT_IMAGE", "1")
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = lab_fft(x)
    assert plan.kernel_name(1).startswith("image") and plan.num_launches == 1, plan.kernel_name(1)
    assert plan.device_status() == 0           # no bounded spin expired, no surplus workgroup (sticky device flags)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    again, _ = lab_fft(x)
    assert np.array_equal(again, out)          # barrier counters are reset per launch; results are reproducible
    back, _ = lab_fft(out, inverse=True)
    assert rel_l2(back, x) < REL_L2_TOL_F32


@pytest.mark.parametrize("shape", [(1, 7680, 64), (2, 5000, 24), (1, 8192, 3, 5)])
def test_long_strided_dimension_through_transposes(shape, monkeypatch):
    """The fallback for long strided dimensions (MIFFT_FOURSTEP_STRIDED=0, or a factor without a fused column tile): the
    reference's own route -- transpose -> row kernel -> transpose through the plan scratch."""
    monkeypatch.setenv("MIFFT_FOURSTEP_STRIDED", "0")
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = lab_fft(x)
    assert plan.kernel_name(0) == "transpose"
    truth = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32
    back, _ = lab_fft(out, inverse=True)
    assert rel_l2(back, x) < REL_L2_TOL_F32


def test_scratch_allocation_failure_is_an_error_not_a_fallback(monkeypatch):
    """A route that needs the plan scratch must report MIFFT_ERR_HIP when the device is out of memory -- not keep
    half-built passes with a NULL scratch, and not hide the failure behind a slower kernel."""
    mf = lab()
    monkeypatch.setenv("MIFFT_TEST_FAIL_SCRATCH_ALLOC", "1")
    for fs in ("1", "0"):
        monkeypatch.setenv("MIFFT_FOURSTEP_STRIDED", fs)
        with pytest.raises(mf.MifftError) as e:
            mf.plan_fft(torch.float32, torch.float32, (1, 7680, 64, 2), (1, 7680, 64, 2), ctx=mf.DeviceContext(0))
        assert e.value.status == -11 and "device allocation" in e.value.message
    monkeypatch.delenv("MIFFT_TEST_FAIL_SCRATCH_ALLOC")
    monkeypatch.delenv("MIFFT_FOURSTEP_STRIDED")
    plan = mf.plan_fft(torch.float32, torch.float32, (1, 7680, 64, 2), (1, 7680, 64, 2), ctx=mf.DeviceContext(0))
    assert plan.scratch_bytes == 7680 * 64 * 8 and plan.num_launches == 3


@pytest.mark.parametrize("shape,inverse", [((1, 93), False), ((16, 93), False), ((37, 93), False), ((301, 93), True),
                                            ((3, 5, 93), False)])
def test_dpp_radix3_rows(shape, inverse, monkeypatch):
    """MIFFT_DPP=1: 93 = 31 * 3 with the radix-3 stage across three adjacent lanes through DPP row shifts (no LDS
    exchange, no workgroup barrier; kernels_dpp.hip).  A negative result kept in the lab build: it measures slower than
    the tile kernel.  Its arithmetic differs from the tile kernel's (one output per lane), so parity is against the
    oracle, and a slab must still equal the same rows of the whole batch bit for bit."""
    monkeypatch.setenv("MIFFT_DPP", "1")
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out, plan = lab_fft(x, inverse=inverse)
    assert plan.kernel_name(len(shape) - 2).startswith("rows93_31x3_dpp")
    assert not np.isnan(out).any()
    assert rel_l2(out, O.fftn(x, inverse=inverse)) < REL_L2_TOL_F32
    if shape[0] >= 16:
        part, _ = lab_fft(x, inverse=inverse, first=5, count=7)
        assert np.array_equal(part[5:12], out[5:12]) and np.isnan(part[:5]).all() and np.isnan(part[12:]).all()


@pytest.mark.parametrize("shape,dtype", [((3, 640, 480), np.float32), ((2, 480, 640), np.float32), ((2, 128, 128, 128), np.float32),
                                         ((3, 64, 64, 64), np.float32),
                                         ((4, 640, 50), np.float32),      # ragged last column tile
                                         ((3, 270, 333), np.float32),     # odd lengths: no self-mirrored middle column
                                         ((2, 12, 10, 21), np.float32),   # tiles across the rows of the trailing dimensions
                                         ((1, 1920, 200), np.float32),    # 8-column tiles: no carried column
                                         ((2, 360, 280), np.float64), ((2, 20, 24, 28), np.float64),
                                         ((1, 64, 16, 8, 64), np.float32), ((2, 12, 5, 6, 14), np.float32),  # three trailing dimensions
                                         ((400, 640, 48), np.float32),    # runs of several tiles per workgroup across images
                                         # three passes (no plane fits LDS): half-store ROW pass, middle pass over half the columns,
                                         # last pass halving the last dimension
                                         ((1, 12, 200, 180), np.float32), ((2, 10, 130, 150), np.float32), ((1, 8, 120, 100), np.float64),
                                         ((1, 16, 256, 256), np.float32), ((3, 6, 96, 250), np.float32)])
@pytest.mark.parametrize("inverse", [False, True])
def test_hermitian_twins_forced_on_small_shapes(shape, dtype, inverse, monkeypatch):
    """MIFFT_HERM=2 takes the Hermitian twin of the last pass wherever one exists, also where the plan-time policy of the
    product (herm_pays: fewer rounds of the persistent grid, whole-line tiles or a cache-resident tensor) would not: the
    ragged, odd, tiny and row-crossing geometries of the mirrored stores and of the carried column."""
    monkeypatch.setenv("MIFFT_HERM", "2")

    def fft_fn(x, *, inverse, out_dtype):
        assert x.dtype == out_dtype
        return lab_fft(x, inverse=inverse)

    small = np.prod(shape) <= 400000
    check_hermitian_plan(fft_fn, shape, dtype, inverse, oracle=O.fftn if small else None)


@pytest.mark.parametrize("shape,dtype", [((3, 400, 96), np.float32), ((2, 300, 250), np.float32), ((2, 240, 96), np.float64),
                                         ((1, 12, 200, 180), np.float32), ((25, 640, 480), np.float32)])  # (no plane fits LDS)
@pytest.mark.parametrize("inverse", [False, True])
def test_packed_real_rows_opt_in(shape, dtype, inverse, monkeypatch):
    """MIFFT_R2C=1: the first pass of a real-input plan with a Hermitian last pass reads its rows of N reals as N / 2 packed
    complex points and unpacks the half spectrum in its store loop (TileCfg::R2C).  Parity only: it is no faster than the tuned
    half-store row kernels (DESIGN_EXPERIMENTS.md R3.8)."""
    monkeypatch.setenv("MIFFT_HERM", "2")
    monkeypatch.setenv("MIFFT_R2C", "1")
    seen = []

    def fft_fn(x, *, inverse, out_dtype):
        out, plan = lab_fft(x, inverse=inverse)
        seen.append([plan.kernel_name(d) for d in range(len(shape) - 1)])
        return out, plan

    check_hermitian_plan(fft_fn, shape, dtype, inverse, oracle=O.fftn if np.prod(shape) <= 400000 else None)
    assert "_r2c_" in seen[0][-1], seen[0]


def test_packed_real_rows_widen_integer_input(monkeypatch):
    """... and with uint8 input: pairs of bytes widened as the packed row is loaded."""
    monkeypatch.setenv("MIFFT_HERM", "2")
    monkeypatch.setenv("MIFFT_R2C", "1")
    mf = lab()
    rng = np.random.default_rng(5)
    shape = (3, 360, 64)
    a = rng.integers(0, 255, size=shape + (1,)).astype(np.uint8)
    x = torch.from_numpy(a).to("cuda:0")
    out = torch.full(shape + (2,), float("nan"), dtype=torch.float32, device="cuda:0")
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(x.dtype, torch.float32, tuple(x.shape), tuple(out.shape), ctx=ctx)
        mf.fft(out, x, ctx, plan=plan)
        ctx.synchronize()
    assert "_r2c_" in plan.kernel_name(1) and plan.kernel_name(0).endswith(("_h", "_h_jit")), [plan.kernel_name(0), plan.kernel_name(1)]
    truth = np.fft.fftn(a[..., 0].astype(np.float64), axes=(1, 2))
    assert rel_l2(out.cpu().numpy(), from_complex(truth, np.float64)) < REL_L2_TOL_F32
