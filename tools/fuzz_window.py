"""Randomised check of batched 1-D transforms whose tensors sit in and around the store-policy window (0.1 - 0.9 GB moved
per exec): arbitrary lengths, fp32 / fp64, real / complex, forward / inverse; sampled rows against fp64 numpy.
    python tools/fuzz_window.py [cases] [seed]"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathon_fft_amd as mf  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = mf.DeviceContext(0)
fails, fam = 0, collections.Counter()
for i in range(cases):
    n = int(rng.choice([rng.integers(8, 300), rng.integers(8, 4097), 2 ** int(rng.integers(3, 13))]))
    f64 = rng.random() < 0.3
    comps = 1 if rng.random() < 0.3 else 2
    inverse = rng.random() < 0.3
    total = float(rng.choice([0.12e9, 0.3e9, 0.45e9, 0.58e9, 0.8e9]))
    esz = 16 if f64 else 8
    batch = max(1, int(total / (2 * esz) / n))
    dt = torch.float64 if f64 else torch.float32
    try:
        x = torch.randn((batch, n, comps), device="cuda:0", dtype=dt)
        out = torch.full((batch, n, 2), float("nan"), device="cuda:0", dtype=dt)
        plan = mf.plan_fft(dt, dt, x.shape, out.shape, inverse=inverse, ctx=ctx)
    except mf.MifftError as e:
        if e.status in (-5, -7):   # a prime factor the reference's default GPU bases (2..32) do not cover
            continue
        raise
    mf.fft(out, x, ctx, plan=plan)
    ctx.synchronize()
    name = plan.kernel_name(0)
    fam["nts" if "_nts" in name else "nt" if name.endswith(("_nt", "_nt_jit")) else "plain"] += 1
    idx = sorted({0, 1, batch // 2, batch - 1, ((batch - 1) // 64) * 64})
    sel = torch.tensor(idx, device="cuda:0")
    xs = x.index_select(0, sel).cpu().numpy().astype(np.float64)
    xc = xs[..., 0] + (1j * xs[..., 1] if comps == 2 else 0)
    ref = np.fft.ifft(xc, axis=1) if inverse else np.fft.fft(xc, axis=1)
    got = out.index_select(0, sel).cpu().numpy().astype(np.float64)
    gc = got[..., 0] + 1j * got[..., 1]
    err = np.abs(gc - ref).max() / max(np.abs(ref).max(), 1e-30)
    nan = bool(torch.isnan(out).any())
    tol = 1e-11 if f64 else 3e-5
    if err > tol or nan:
        fails += 1
        print("FAIL", n, batch, dt, comps, inverse, name, err, nan, flush=True)
    del x, out, plan
print(f"{cases} cases, {fails} failures; kernels: {dict(fam)}")
sys.exit(1 if fails else 0)
