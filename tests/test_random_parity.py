"""Randomised parity sweeps.

CPU (not gpu): hypothesis drives the oracle over random shapes / user radix lists against fp64 pocketfft.
GPU: a seeded sweep of (shape, bases, dtype, direction, input kind, kernel family) against the oracle."""
import itertools

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import REL_L2_TOL_F32, REL_L2_TOL_F64, from_complex, rel_l2, to_complex
from oracle import mifft_oracle as O

SMALL_PRIMES = [2, 3, 5, 7, 11, 13]


@st.composite
def dim_with_bases(draw):
    """a length and a user `bases` list that factors it (possibly unsorted / composite / incomplete)"""
    factors = draw(st.lists(st.sampled_from(SMALL_PRIMES), min_size=1, max_size=5))
    n = int(np.prod(factors))
    if n > 2000:
        factors = factors[:3]
        n = int(np.prod(factors))
    style = draw(st.integers(0, 2))
    if style == 0:          # the exact factor list, shuffled
        bases = list(draw(st.permutations(factors)))
    elif style == 1:        # distinct primes only: the planner repeats them
        bases = sorted(set(factors))
    else:                   # merge two factors into a composite base
        f = list(factors)
        if len(f) >= 2:
            f = [f[0] * f[1]] + f[2:]
        bases = f
    return n, bases


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.lists(dim_with_bases(), min_size=1, max_size=3), st.integers(1, 3), st.booleans(), st.booleans())
def test_oracle_vs_pocketfft_random(dims, batch, inverse, real_input):
    shape = tuple(d[0] for d in dims)
    bases = [d[1] for d in dims]
    if int(np.prod(shape)) > 40000:
        return
    rng = np.random.default_rng(abs(hash((shape, batch, inverse))) % (2 ** 32))
    comps = 1 if (real_input and not inverse) else 2
    x = rng.standard_normal((batch,) + shape + (comps,))
    try:
        out = O.fftn(x, inverse=inverse, bases=bases)
    except O.OracleError as e:       # the reference rejects some lists too (greedy over-shoot), e.g. [4, 2] for 32
        assert e.status == -5
        return
    xc = x[..., 0] if comps == 1 else to_complex(x)
    axes = tuple(range(1, len(shape) + 1))
    ref = np.fft.ifftn(xc, axes=axes) if inverse else np.fft.fftn(xc, axes=axes)
    assert rel_l2(out, from_complex(ref, np.float64)) < 1e-12


# ---------------------------------------------------------------------------------------------
GPU_LENGTHS = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16, 20, 21, 24, 30, 31, 32, 35, 48, 60, 64, 93, 97, 100, 120, 128,
               144, 200, 243, 256, 360, 480, 500, 512, 625, 640, 1000, 1024, 1331, 2048, 4096]


def _cases():
    rng = np.random.default_rng(20260504)
    cases = []
    for i in range(72):
        nd = int(rng.choice([1, 1, 1, 2, 2, 3]))
        pool = GPU_LENGTHS if nd == 1 else [n for n in GPU_LENGTHS if n <= (640 if nd == 2 else 64)]
        shape = tuple(int(rng.choice(pool)) for _ in range(nd))
        batch = int(rng.integers(1, 40 if nd == 1 else 4))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        inverse = bool(rng.random() < 0.35)
        real = bool(rng.random() < 0.3) and not inverse
        faithful = bool(rng.random() < 0.4)
        cases.append(((batch,) + shape, dtype, inverse, real, faithful))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype,inverse,real,faithful", _cases())
def test_gpu_vs_oracle_random(shape, dtype, inverse, real, faithful):
    import torch
    import hackathon_fft_amd as mf
    rng = np.random.default_rng(abs(hash(shape)) % (2 ** 32))
    x = rng.standard_normal(shape + ((1,) if real else (2,))).astype(dtype)
    ref = O.fftn(x, inverse=inverse, out_dtype=dtype)      # reference default bases (CPU list) -- same DFT
    xd = torch.from_numpy(x).to("cuda:0")
    out = torch.full(shape + (2,), float("nan"), dtype=xd.dtype, device="cuda:0")
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(xd.dtype, xd.dtype, xd.shape, out.shape, inverse=inverse, faithful_stages=faithful, ctx=ctx)
    mf.fft(out, xd, ctx, plan=plan)
    ctx.synchronize()
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    tol = REL_L2_TOL_F32 if dtype == np.float32 else REL_L2_TOL_F64
    names = [plan.kernel_name(d) for d in range(len(shape) - 1)]
    assert rel_l2(got, ref) < tol, (shape, dtype, inverse, real, faithful, names)
    if faithful:
        assert all(n == "generic" for n in names)
