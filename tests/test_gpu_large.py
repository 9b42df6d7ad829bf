"""Tensors beyond 2^31 complex elements (17 GB each; the part has 288 GB): every offset on the path must be 64-bit clean.
No CPU reference at this size -- sampled transforms (first, last, and the ones on either side of the 2^31-element and
2^32-byte marks) are compared with the same inputs transformed as a small batch, every output element must have been
written (NaN pre-fill), and Parseval must hold on the sampled transforms."""
import numpy as np
import pytest
import torch

import hackathon_fft_amd as mf

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)

CASES = [
    ((2_200_000, 1024), [[2]]),          # streaming row kernel, 2.25e9 elements
    ((7_500, 640, 480), None),           # row pass + in-place column tiles
    ((1_100, 128, 128, 128), None),      # fused planes + wide column tiles
    ((2_100, 1 << 20), None),            # four-step on the contiguous dimension (+ scratch for the transposed store or none)
    ((70, 7680, 4320), None),            # strided four-step through the plan scratch
]


@pytest.mark.parametrize("shape,bases", CASES)
def test_tensors_beyond_2_31_elements(shape, bases):
    free, _ = torch.cuda.mem_get_info(DEV)
    per = int(np.prod(shape)) * 8
    if free < 3.6 * per:
        pytest.skip(f"needs {3.6 * per / 1e9:.0f} GB of device memory")
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(shape + (2,), generator=g, device=DEV, dtype=torch.float32)
    out = torch.full_like(x, float("nan"))
    ctx = mf.DeviceContext(0)
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=bases, ctx=ctx)
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    b, n_per = shape[0], int(np.prod(shape[1:]))
    assert b * n_per > 2 ** 31
    # transforms around the 2^32-byte, 2^31-element and 2^32-element (x2 floats) marks, plus the ends
    marks = [2 ** 32 // 8 // n_per, 2 ** 31 // n_per, 2 ** 31 // 2 // n_per]
    idx = sorted({0, 1, b // 2, b - 2, b - 1} | {m + d for m in marks for d in (-1, 0, 1) if 0 <= m + d < b})
    sel = torch.tensor(idx, device=DEV)
    xs = x.index_select(0, sel).contiguous()
    small = mf.plan_fft(torch.float32, torch.float32, xs.shape, xs.shape, bases=bases, ctx=ctx)
    ref = torch.empty_like(xs)
    mf.fft(ref, xs, ctx, plan=small)
    ctx.synchronize()
    got = out.index_select(0, sel)
    err = ((got.double() - ref.double()).reshape(len(idx), -1).norm(dim=1) / ref.double().reshape(len(idx), -1).norm(dim=1))
    assert err.max().item() < 2e-6, (shape, [i for i, e in zip(idx, err.tolist()) if e >= 2e-6])
    ex = (xs.double() ** 2).reshape(len(idx), -1).sum(1)
    eX = (got.double() ** 2).reshape(len(idx), -1).sum(1)
    assert ((eX / (n_per * ex) - 1).abs().max().item()) < 1e-5
    # every element written: chunked so that the check itself stays small
    flat = out.reshape(-1)
    step = 1 << 30
    for s in range(0, flat.numel(), step):
        assert not torch.isnan(flat[s:s + step]).any(), (shape, s)
    del x, out, flat
    torch.cuda.empty_cache()
