"""L2-resident image kernel (rows, XCD-local barrier, columns from L2) against separate row and column passes
(default).  GPU box:  MIFFT_JIT_IMAGE=1 python tools/image_probe.py ; python tools/image_probe.py"""
import os, sys
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hackathon_fft_amd as mf

for shape in ((100, 640, 480), (128, 512, 512), (400, 256, 256), (64, 600, 500), (37, 640, 480)):
    x = torch.randn(shape + (2,), device="cuda:0")
    out = torch.full_like(x, float("nan"))
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan)
        ctx.synchronize()
        got = torch.view_as_complex(out[:3].contiguous()).cpu().numpy()
        ref = np.fft.fftn(torch.view_as_complex(x[:3].contiguous()).cpu().numpy().astype(np.complex128), axes=(1, 2))
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        last = torch.view_as_complex(out[-1:].contiguous()).cpu().numpy()
        refl = np.fft.fftn(torch.view_as_complex(x[-1:].contiguous()).cpu().numpy().astype(np.complex128), axes=(1, 2))
        errl = np.linalg.norm(last - refl) / np.linalg.norm(refl)
        mf.time_fft(out, x, plan=plan, iters=5, ctx=ctx)
        ms = mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx)
        print(f"{shape}: {ms:.4f} ms  launches {plan.num_launches}  {plan.kernel_name(1)}  rel err first/last {err:.2e} {errl:.2e}")
