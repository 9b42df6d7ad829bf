"""The CPU oracle against the reference's golden vectors and fp64 pocketfft.

Reproduces the reference's 1-D (N, bases) matrix (fft/tests.mojo:274-371) in both
directions, the 2-D and 3-D uint8 cases (fft/tests.mojo:461-518, 908-970), and adds what
the reference lacks: fp32 at the BASELINE shapes.
"""
import numpy as np
import pytest

from conftest import REF_ATOL, REF_RTOL, from_complex, load_matrix, rel_l2, to_complex
from oracle import mifft_oracle as O


@pytest.mark.parametrize("n,bases", load_matrix())
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_1d_forward_real_input(golden_1d, n, bases, dtype):
    pairs = golden_1d["values"][str(n)]
    x = np.array([p["x"] for p in pairs], dtype=dtype).reshape(len(pairs), n, 1)
    expected = np.array([p["X"] for p in pairs])
    out = O.fftn(x, bases=[list(bases)])
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, expected, atol=REF_ATOL, rtol=REF_RTOL)
    ref = np.fft.fft(x[..., 0].astype(np.float64))
    tol = 1e-12 if dtype == np.float64 else 2e-6
    assert rel_l2(out, from_complex(ref, np.float64)) < tol


@pytest.mark.parametrize("n,bases", load_matrix())
def test_1d_inverse_complex_input(golden_1d, n, bases):
    pairs = golden_1d["values"][str(n)]
    spectrum = np.array([p["X"] for p in pairs], dtype=np.float64)
    series = np.array([p["x"] for p in pairs], dtype=np.float64)
    out = O.fftn(spectrum, inverse=True, bases=[list(bases)])
    np.testing.assert_allclose(out[..., 0], series, atol=REF_ATOL, rtol=REF_RTOL)
    np.testing.assert_allclose(out[..., 1], 0, atol=REF_ATOL, rtol=REF_RTOL)


def test_unused_length_30_vectors(golden_1d):
    # _get_test_values_30 exists in the reference but no test consumes it (fft/_test_values.mojo:552)
    pairs = golden_1d["values"]["30"]
    x = np.array([p["x"] for p in pairs], dtype=np.float64).reshape(len(pairs), 30, 1)
    out = O.fftn(x, bases=[[5, 3, 2]])
    np.testing.assert_allclose(out, np.array([p["X"] for p in pairs]), atol=REF_ATOL, rtol=REF_RTOL)


@pytest.mark.parametrize("which,axes", [("2d", (1, 2)), ("3d", (1, 2, 3))])
def test_nd_uint8_default_bases(golden_2d, golden_3d, which, axes):
    g = golden_2d if which == "2d" else golden_3d
    x = np.array(g["x"], dtype=np.uint8)[None, ..., None]
    out = O.fftn(x, out_dtype=np.float64)
    expected = np.array(g["X_flat"]).reshape(out.shape)
    np.testing.assert_allclose(out, expected, atol=REF_ATOL, rtol=REF_RTOL)
    ref = np.fft.fftn(x[..., 0].astype(np.float64), axes=axes)
    assert rel_l2(out, from_complex(ref, np.float64)) < 1e-13
    back = O.fftn(out, inverse=True)
    np.testing.assert_allclose(back[..., 0], x[..., 0], atol=1e-9)
    np.testing.assert_allclose(back[..., 1], 0, atol=1e-9)


BASELINE_SMALL = [
    ((6, 128), None),            # config 1 shape (500k x 128), sub-batch
    ((4, 1024), [[2]]),          # config 2, radix-2 Stockham
    ((6, 93), [[31, 3]]),        # config 3, generic-prime radix 31
    ((1, 640, 480), None),       # config 4
    ((1, 32, 32, 32), None),     # config 5 at reduced size
]


@pytest.mark.parametrize("shape,bases", BASELINE_SMALL)
def test_fp32_baseline_shapes_vs_pocketfft(shape, bases):
    rng = np.random.default_rng(1234)
    x = rng.standard_normal(shape + (2,)).astype(np.float32)
    out = O.fftn(x, bases=bases)
    ref = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(ref, np.float64)) < 2e-6
    # round trip through the inverse
    back = O.fftn(out, inverse=True, bases=bases)
    assert rel_l2(back, x) < 2e-6


@pytest.mark.parametrize("shape", [(2, 4, 6, 5, 8), (1, 3, 4, 2, 5, 6)])
def test_rank_above_three_like_the_reference_allows(shape):
    # the reference's layout check only asks for rank > 2 (fft/fft/fft.mojo:22-26); its bench lists 4-D shapes
    rng = np.random.default_rng(8)
    x = rng.standard_normal(shape + (2,))
    out = O.fftn(x)
    ref = np.fft.fftn(to_complex(x), axes=tuple(range(1, len(shape))))
    assert rel_l2(out, from_complex(ref, np.float64)) < 1e-13


def test_threads_do_not_change_results():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 12, 10, 2)).astype(np.float32)
    a = O.fftn(x, cpu_workers=1)
    b = O.fftn(x, cpu_workers=4)
    assert np.array_equal(a, b)


def test_every_output_element_written_and_x_untouched():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 6, 4, 8, 2))
    x0 = x.copy()
    plan = O.plan_fft(np.float64, np.float64, x.shape, x.shape)
    out = np.full(x.shape, np.nan)
    O.fft(out, x, plan=plan)
    assert not np.isnan(out).any() and np.array_equal(x, x0)
    # partial batch range leaves the rest untouched
    out2 = np.full(x.shape, np.nan)
    O.fft(out2, x, plan=plan, first=1, count=1)
    assert np.isnan(out2[0]).all() and np.isnan(out2[2]).all() and np.array_equal(out2[1], out[1])


@pytest.mark.parametrize("kind", ["int8", "int16", "uint16", "float16", "bfloat16"])
@pytest.mark.parametrize("comps", [1, 2])
def test_oracle_widens_narrow_input_types_exactly(kind, comps):
    """`x.load(...).cast[out_dtype]()` (fft/fft/_fft.mojo:243-257) for the element types the GPU path accepts beyond the
    reference's own uint8 tests: the oracle's result on the narrow tensor equals its result on the same values held in
    float64 (every such value is exact in float64), bit for bit."""
    rng = np.random.default_rng(len(kind) + comps)
    shape = (3, 12, 10, comps)
    in_dtype = None
    if kind == "bfloat16":
        f = rng.standard_normal(shape).astype(np.float32)
        bits = (f.view(np.uint32) >> 16).astype(np.uint16)          # truncate to bfloat16
        vals = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        a, in_dtype = bits, O.BF16
    elif kind == "float16":
        a = (rng.standard_normal(shape) * 4).astype(np.float16)
        vals = a.astype(np.float64)
    else:
        info = np.iinfo(kind)
        a = rng.integers(info.min, info.max + 1, size=shape).astype(kind)
        vals = a.astype(np.float64)
    got = O.fftn(a, out_dtype=np.float64, in_dtype=in_dtype)
    want = O.fftn(vals, out_dtype=np.float64)
    assert np.array_equal(got, want)
    z = vals[..., 0] + (1j * vals[..., 1] if comps == 2 else 0)
    truth = np.fft.fftn(z, axes=(1, 2))
    assert np.abs(got[..., 0] + 1j * got[..., 1] - truth).max() <= 1e-11 * np.abs(truth).max()


def test_oracle_half_conversion_is_exact_for_every_finite_value():
    h = np.arange(0, 65536, dtype=np.uint16).view(np.float16)
    h = h[np.isfinite(h.astype(np.float32))]
    x = h[:len(h) // 2 * 2].reshape(-1, 2, 1)
    y = O.fftn(x, out_dtype=np.float64)          # 2-point DFT of (a, b): (a + b, a - b), exact in float64
    a, b = x[:, 0, 0].astype(np.float64), x[:, 1, 0].astype(np.float64)
    assert np.array_equal(y[:, 0, 0], a + b) and np.array_equal(y[:, 1, 0], a - b)
