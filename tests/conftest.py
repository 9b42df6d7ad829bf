import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tolerance of the reference's own tests (fft/tests.mojo:40-41)
REF_ATOL, REF_RTOL = 1e-2, 1e-5
# BASELINE.json north_star: <= 1e-5 relative error vs the reference CPU path,
# measured per transform as ||y - y_ref||_2 / ||y_ref||_2 ("Tolerance definition")
REL_L2_TOL_F32 = 1e-5
REL_L2_TOL_F64 = 1e-12


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_1d():
    with open(os.path.join(GOLDEN, "fft_1d.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_2d():
    with open(os.path.join(GOLDEN, "fft_2d.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_3d():
    with open(os.path.join(GOLDEN, "fft_3d.json")) as f:
        return json.load(f)


def load_matrix():
    with open(os.path.join(GOLDEN, "fft_1d.json")) as f:
        d = json.load(f)
    return [(m["n"], tuple(m["bases"])) for m in d["matrix"]]


def rel_l2(y, ref):
    """max over transforms (leading axis) of ||y - ref||_2 / ||ref||_2; inputs interleaved (.., 2)."""
    y = np.asarray(y, dtype=np.float64).reshape(y.shape[0], -1)
    ref = np.asarray(ref, dtype=np.float64).reshape(ref.shape[0], -1)
    num = np.linalg.norm(y - ref, axis=1)
    den = np.linalg.norm(ref, axis=1)
    den[den == 0] = 1.0
    return float((num / den).max())


def to_complex(a):
    a = np.asarray(a)
    return a[..., 0].astype(np.float64) + 1j * a[..., 1].astype(np.float64)


def from_complex(c, dtype):
    return np.stack([c.real, c.imag], axis=-1).astype(dtype)


def check_hermitian_plan(fft_fn, shape, dtype, inverse, oracle=None):
    """Real input through a plan whose LAST pass is a Hermitian twin (kernel name `..._h`) behind a half-store pass
    (`..._hs`): parity with fp64 pocketfft (and
    the oracle when given), Y[-k, -c] == conj(Y[k, c]) BIT FOR BIT wherever one of the two was stored as the conjugate of the
    other, and agreement to rounding with the complex-input plan on (x, 0), which runs the ordinary kernels.
    fft_fn(x, inverse=, out_dtype=) -> (out, plan)."""
    rng = np.random.default_rng(sum(shape) + int(inverse))
    x = rng.standard_normal(shape + (1,)).astype(dtype)
    out, plan = fft_fn(x, inverse=inverse, out_dtype=dtype)
    assert plan.kernel_name(0).endswith(("_h", "_h_jit")), plan.kernel_name(0)
    # ... and a pass before it stores only the half of its dimension that the last pass reads (TileCfg::HS): the pass right
    # before the last, or the first of three
    names = [plan.kernel_name(d) for d in range(len(shape) - 1)]
    assert any("_hs" in n for n in names[1:]), names
    assert not np.isnan(out).any()
    axes = tuple(range(1, len(shape)))
    z = x[..., 0].astype(np.float64)
    truth = np.fft.ifftn(z, axes=axes) if inverse else np.fft.fftn(z, axes=axes)
    assert rel_l2(out, from_complex(truth, np.float64)) < (REL_L2_TOL_F32 if dtype == np.float32 else 1e-11)
    del truth, z
    if oracle is not None:
        assert rel_l2(out, oracle(x, inverse=inverse, out_dtype=dtype)) < (REL_L2_TOL_F32 if dtype == np.float32 else REL_L2_TOL_F64)
    # a column c (index over the trailing dimensions) that is its own mirror image is transformed on its own: Hermitian along
    # k to rounding only
    y = out[..., 0] + 1j * out[..., 1]
    mirrored = y
    self_col = np.ones(shape[2:], dtype=bool)
    for ax in axes:
        mirrored = np.roll(np.flip(mirrored, axis=ax), 1, axis=ax)
        if ax >= 2:
            n = shape[ax]
            k = np.arange(n)
            self_col = self_col & (k == (n - k) % n).reshape([-1 if a == ax - 2 else 1 for a in range(len(shape) - 2)])
    assert np.array_equal(np.conj(mirrored)[:, :, ~self_col], y[:, :, ~self_col])
    assert np.abs(np.conj(mirrored) - y).max() <= 2e-5 * np.abs(y).max()
    del mirrored, y
    xc = np.concatenate([x, np.zeros_like(x)], axis=-1)
    outc, planc = fft_fn(xc, inverse=inverse, out_dtype=dtype)
    assert not planc.kernel_name(0).endswith(("_h", "_h_jit"))
    assert rel_l2(out, outc) < (2e-6 if dtype == np.float32 else 1e-12)
