"""Generated table entries against the runtime-specialised kernels of the same lengths (lab build, MIFFT_SKIP_GEN_TABLE = 0 / 1 in
one process), complex input, whole transform in ms.
    MIFFT_LIBRARY=hackathon_fft_amd/csrc/libmifft_lab.so python tools/gen_vs_jit_probe.py [shape ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hackathon_fft_amd as mf  # noqa: E402

SHAPES = ["10x1920x1080", "10x1080x1920", "1x3840x2160", "1x2160x3840", "40x1920x1080", "4x3840x2160", "100x600x500", "20x1000x1000",
          "50x720x1280", "200x360x360", "30x1200x800", "1x360x360x360", "4x200x200x200", "2x320x320x320", "100x96x96x96",
          "500000x96", "200000x200", "100000x360", "50000x1000", "20000x2000", "300000x100", "100000x243", "60000x625"]


def main():
    shapes = [a for a in sys.argv[1:] if "x" in a] or SHAPES
    print(f"{'shape':>18} {'table ms':>9} {'jit ms':>9} {'jit/table':>9}  table kernels | jit kernels")
    for spec in shapes:
        shape = tuple(int(v) for v in spec.split("x"))
        x = torch.randn(shape + (2,), device="cuda:0")
        out = torch.empty(shape + (2,), device="cuda:0")
        res, names = {}, {}
        for mode in ("0", "1", "0", "1"):
            os.environ["MIFFT_SKIP_GEN_TABLE"] = mode
            with mf.DeviceContext(0) as ctx:
                plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
                mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
                ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
                res[mode] = min(ms, res.get(mode, 1e9))
                names[mode] = [plan.kernel_name(d) for d in range(len(shape) - 1)]
            del plan
        print(f"{spec:>18} {res['0']:9.4f} {res['1']:9.4f} {res['1'] / res['0']:9.3f}  {names['0']} | {names['1']}", flush=True)
        del x, out


if __name__ == "__main__":
    main()
