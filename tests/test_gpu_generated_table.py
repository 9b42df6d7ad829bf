"""Every fused-kernel configuration emitted by tools/gen_fast_table.py, contiguous and strided,
against fp64 pocketfft (the oracle is pinned to pocketfft at <= 6e-13 in test_oracle_golden.py and
would take minutes at the larger lengths)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import hackathon_fft_amd as mf
from conftest import REL_L2_TOL_F32, ROOT, from_complex, rel_l2, to_complex

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("gen_fast_table", os.path.join(ROOT, "tools", "gen_fast_table.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)


def _run(x_np):
    x = torch.from_numpy(x_np).to("cuda:0")
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(x.dtype, x.dtype, x.shape, x.shape, ctx=ctx)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    return out.cpu().numpy(), plan


@pytest.mark.parametrize("n", _gen.SIZES)
def test_rows(n):
    rng = np.random.default_rng(n)
    batch = 131 if n <= 512 else 7      # ragged against every tile size
    x = rng.standard_normal((batch, n, 2)).astype(np.float32)
    out, plan = _run(x)
    assert not np.isnan(out).any()
    truth = np.fft.fft(to_complex(x), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32, plan.kernel_name(0)
    assert plan.kernel_name(0) != "generic", n


@pytest.mark.parametrize("n", [s for s in _gen.SIZES if s <= 4096])
def test_cols(n):
    rng = np.random.default_rng(n + 1)
    inner = 40                          # not a multiple of the 16-column tile: ragged last tile
    x = rng.standard_normal((2, n, inner, 2)).astype(np.float32)
    out, plan = _run(x)
    assert not np.isnan(out).any()
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32, plan.kernel_name(0)
    assert plan.kernel_name(0) != "generic", n


@pytest.mark.parametrize("n", _gen.SIZES_F64)
def test_rows_f64(n):
    rng = np.random.default_rng(n + 2)
    x = rng.standard_normal((67 if n <= 512 else 5, n, 2))
    out, plan = _run(x)
    truth = np.fft.fft(to_complex(x), axis=1)
    assert rel_l2(out, from_complex(truth, np.float64)) < 1e-12, plan.kernel_name(0)
    assert plan.kernel_name(0) != "generic", n


@pytest.mark.parametrize("n", [s for s in _gen.SIZES_F64 if s <= 2048])
def test_cols_f64(n):
    rng = np.random.default_rng(n + 3)
    x = rng.standard_normal((2, n, 20, 2))     # 20 columns: ragged against the 8-column f64 tile
    out, plan = _run(x)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < 1e-12, plan.kernel_name(0)
    assert plan.kernel_name(0) != "generic", n


@pytest.mark.parametrize("n", [s for s in _gen.SIZES if s <= 128])
@pytest.mark.parametrize("inner,width", [(64, 64), (96, 32)])
def test_wide_cols(n, inner, width):
    """strides that are a multiple of 32 / 64 columns take the wide tiles (256- / 512-byte runs)"""
    rng = np.random.default_rng(n + inner)
    x = rng.standard_normal((3, n, inner, 2)).astype(np.float32)
    out, plan = _run(x)
    name = plan.kernel_name(0)
    if n * width <= 4096 or (width == 64 and n * 32 <= 4096):
        assert "_w" in name or name.startswith("plane"), name   # (small n x inner planes are fused instead)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < REL_L2_TOL_F32, name


@pytest.mark.parametrize("n", [s for s in _gen.SIZES_F64 if s <= 128])
def test_wide_cols_f64(n):
    rng = np.random.default_rng(n + 5)
    x = rng.standard_normal((2, n, 64, 2))
    out, plan = _run(x)
    truth = np.fft.fftn(to_complex(x), axes=(1, 2))
    assert rel_l2(out, from_complex(truth, np.float64)) < 1e-12, plan.kernel_name(0)
