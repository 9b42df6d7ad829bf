"""Edge geometries of the tile scheduler: single transforms, ragged last tiles, inner dimensions narrower
than a column tile, very large batches of tiny transforms, and plans running concurrently on two streams."""
import numpy as np
import pytest
import torch

import hackathon_fft_amd as mf
from conftest import REL_L2_TOL_F32, from_complex, rel_l2, to_complex

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def np_fft(x_slice, axes):
    """fp64 pocketfft (NumPy) of an interleaved device slice -> interleaved float64 tensor on the host.
    (No vendor FFT anywhere, tests included: torch.fft would be rocFFT.)"""
    xn = x_slice.detach().cpu().numpy().astype(np.float64)
    t = np.fft.fftn(xn[..., 0] + 1j * xn[..., 1], axes=axes)
    return torch.from_numpy(np.stack([t.real, t.imag], axis=-1))


def run(x, **kw):
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(x.dtype, x.dtype, x.shape, x.shape, ctx=ctx, **kw)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    return out, plan


def check(x, axes):
    out, plan = run(x)
    xn = x.cpu().numpy()
    truth = np.fft.fftn(to_complex(xn), axes=axes)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    b = xn.shape[0]
    err = rel_l2(got.reshape(b, -1, 2), from_complex(truth, np.float64).reshape(b, -1, 2))
    assert err < REL_L2_TOL_F32, ([plan.kernel_name(d) for d in range(x.dim() - 2)], err)
    return plan


@pytest.mark.parametrize("shape", [(1, 1024), (1, 93), (1, 128), (5, 1024), (63, 93), (65, 93), (17, 128), (1, 480),
                                   (1, 16384), (2, 8192), (1, 64, 64), (3, 64, 64), (1, 640, 480), (1, 128, 128, 128)])
def test_single_and_ragged_batches(shape):
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(shape + (2,), generator=g, device=DEV)
    check(x, tuple(range(1, len(shape))))


@pytest.mark.parametrize("shape", [(3, 64, 5), (2, 128, 17), (2, 640, 33), (4, 256, 16), (2, 12, 128, 3), (1, 480, 1000)])
def test_inner_dimension_vs_column_tile(shape):
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(shape + (2,), generator=g, device=DEV)
    check(x, tuple(range(1, len(shape))))


def test_two_million_tiny_transforms():
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((2_000_000, 16, 2), generator=g, device=DEV)
    out, plan = run(x)
    ref = np_fft(x[:4096], (1,))
    got = out[:4096].double().cpu()
    assert ((got - ref).reshape(4096, -1).norm(dim=1) / ref.reshape(4096, -1).norm(dim=1)).max().item() < 1e-5
    tail = np_fft(x[-7:], (1,))
    assert ((out[-7:].double().cpu() - tail).norm() / tail.norm()).item() < 1e-5
    assert not torch.isnan(out).any()


def test_plans_on_two_streams_do_not_interfere():
    g = torch.Generator(device=DEV).manual_seed(9)
    xa = torch.randn((20000, 1024, 2), generator=g, device=DEV)
    xb = torch.randn((40000, 93, 2), generator=g, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ca, cb = mf.DeviceContext(0, sa), mf.DeviceContext(0, sb)
    pa = mf.plan_fft(torch.float32, torch.float32, xa.shape, xa.shape, bases=[[2]], ctx=ca)
    pb = mf.plan_fft(torch.float32, torch.float32, xb.shape, xb.shape, ctx=cb)
    oa, ob = torch.empty_like(xa), torch.empty_like(xb)
    torch.cuda.synchronize()
    for _ in range(5):
        mf.fft(oa, xa, ca, plan=pa)
        mf.fft(ob, xb, cb, plan=pb)
    ca.synchronize()
    cb.synchronize()
    ra, rb = np_fft(xa[:64], (1,)), np_fft(xb[-64:], (1,))
    assert ((oa[:64].double().cpu() - ra).norm() / ra.norm()).item() < 1e-5
    assert ((ob[-64:].double().cpu() - rb).norm() / rb.norm()).item() < 1e-5


def test_plan_lifecycle():
    x = torch.randn((4, 128, 2), device=DEV)
    for _ in range(50):  # create / run / destroy repeatedly: no leak-induced failure, no stale table
        out, plan = run(x)
        plan.close()
        plan.close()  # idempotent
    ref = np_fft(x, (1,))
    assert ((out.double().cpu() - ref).norm() / ref.norm()).item() < 1e-5


@pytest.mark.parametrize("shape,kw", [((2000, 1024), dict(bases=[[2]])), ((8, 640, 480), {}), ((3, 16384), {}),
                                      ((4, 7680), dict(faithful_stages=True)), ((500, 343), {}),   # runtime-specialised (module launch)
                                      ((2, 1 << 20), {}), ((3, 128, 128), {})])           # four-step, fused plane
def test_exec_is_capturable_into_a_hip_graph(shape, kw):
    """exec enqueues kernels only (no allocation, no attribute call, no sync): it can be captured and replayed."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(shape + (2,), generator=g, device=DEV)
    out = torch.zeros_like(x)
    side = torch.cuda.Stream()
    ctx = mf.DeviceContext(0, side)
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx, **kw)   # big-LDS kernels included
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        mf.fft(out, x, ctx, plan=plan)            # warm-up outside capture
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            mf.fft(out, x, mf.DeviceContext(0), plan=plan)   # current stream = capture stream
    expected = out.clone()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, expected)
    axes = tuple(range(1, len(shape)))
    truth = np_fft(x[:2], axes)
    assert ((out[:2].double().cpu() - truth).norm() / truth.norm()).item() < 1e-5


@pytest.mark.parametrize("n", [93, 128, 256, 512, 1024, 2048, 4096])
def test_streaming_twins_at_streaming_size(n):
    """Above 0.6 GB of traffic per exec the plan picks the non-temporal twin (a different radix sequence for some
    lengths): run it at that size and spot-check rows from the start, the middle and the ragged end."""
    batch = 38_000_000 // n + 3
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn((batch, n, 2), generator=g, device=DEV)
    out, plan = run(x)
    assert plan.kernel_name(0).endswith(("_nt", "_nts")), plan.kernel_name(0)
    for lo in (0, batch // 2, batch - 5):
        ref = np_fft(x[lo:lo + 5], (1,))
        got = out[lo:lo + 5].double().cpu()
        assert ((got - ref).reshape(5, -1).norm(dim=1) / ref.reshape(5, -1).norm(dim=1)).max().item() < 1e-5
    assert not torch.isnan(out).any()


def test_empty_batch_and_bad_slab_ranges():
    ctx = mf.DeviceContext(0)
    for shape in [(0, 1024, 2), (0, 64, 64, 2), (0, 1 << 17, 2)]:
        x = torch.zeros(shape, device=DEV)
        out = torch.zeros(shape, device=DEV)
        plan = mf.plan_fft(torch.float32, torch.float32, shape, shape, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan)      # nothing to do, nothing launched, no error
        ctx.synchronize()
    x = torch.randn((6, 128, 2), device=DEV)
    out = torch.full_like(x, float("nan"))
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
    for first, count in [(-1, 2), (5, 2), (0, 7), (3, -1)]:
        with pytest.raises(mf.MifftError) as e:
            mf.fft(out, x, ctx, plan=plan, first=first, count=count)
        assert e.value.status == -8
    assert torch.isnan(out).all()
    mf.fft(out, x, ctx, plan=plan, first=5, count=1)
    ctx.synchronize()
    assert torch.isnan(out[:5]).all() and not torch.isnan(out[5]).any()


@pytest.mark.parametrize("shape", [(4, 1024), (3, 64, 48), (2, 16, 16, 16), (1, 5120, 8)])
def test_exec_is_graph_capturable(shape):
    """mifft_exec enqueues kernel launches and nothing else (no allocation, no synchronisation, no host-side state that
    changes between execs), so it can be captured into a HIP graph and replayed on new data -- what a host does for the
    launch-bound small-batch regime (one or two 128^3 volumes per GPU in the 8-GPU split).  The last shape runs the
    long-strided four-step through the plan-owned scratch."""
    import hackathon_fft_amd as mf
    rng = np.random.default_rng(sum(shape))
    full = tuple(shape) + (2,)
    x = torch.from_numpy(rng.standard_normal(full).astype(np.float32)).to("cuda:0")
    out = torch.full_like(x, float("nan"))
    side = torch.cuda.Stream()
    ctx = mf.DeviceContext(0, stream=side)
    plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
    with torch.cuda.stream(side):
        mf.fft(out, x, ctx, plan=plan)          # warm-up outside the capture
    side.synchronize()
    ref1 = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        mf.fft(out, x, ctx, plan=plan)
    x.copy_(torch.from_numpy(rng.standard_normal(full).astype(np.float32)))
    out.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    truth = np.fft.fftn(x.cpu().numpy()[..., 0] + 1j * x.cpu().numpy()[..., 1], axes=tuple(range(1, len(shape))))
    got = out.cpu().numpy()
    err = np.linalg.norm(got[..., 0] + 1j * got[..., 1] - truth) / np.linalg.norm(truth)
    assert not np.isnan(got).any() and err < 1e-5
    assert not torch.equal(out, ref1)           # the replay really transformed the new data
