"""MIFFT_JIT=0 (no runtime specialisation) on real-input N-D shapes: the Hermitian / half-store schedules must fall back
consistently when one of their kernels would have needed hipRTC (lab build, GPU box).  python tools/jit0_check.py"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["MIFFT_LIBRARY"] = os.path.join(os.getcwd(), "hackathon_fft_amd/csrc/libmifft_lab.so")
os.environ["MIFFT_JIT"] = "0"
import numpy as np, torch
import hackathon_fft_amd as mf
rng = np.random.default_rng(1)
for shape in [(25, 640, 480), (4, 256, 256, 256), (10, 1920, 1080), (2, 64, 64, 64, 64), (6, 200, 200), (3, 96, 100, 90)]:
    x = torch.from_numpy(rng.standard_normal(shape + (1,)).astype(np.float32)).cuda()
    out = torch.full(shape + (2,), float("nan"), device="cuda")
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(x.dtype, out.dtype, x.shape, out.shape, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
    names = [plan.kernel_name(d) for d in range(len(shape) - 1)]
    z = x.cpu().numpy()[..., 0].astype(np.float64)
    truth = np.fft.fftn(z, axes=tuple(range(1, len(shape))))
    got = out.cpu().numpy().astype(np.float64); g = got[..., 0] + 1j * got[..., 1]
    err = np.linalg.norm(g - truth) / np.linalg.norm(truth)
    print(shape, names, "rel l2 %.2e" % err, "OK" if err < 1e-5 and np.isfinite(got).all() else "FAIL")
