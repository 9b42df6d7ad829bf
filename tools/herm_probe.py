"""Hermitian twins of the last pass of real-input N-D plans (TileCfg::HERM) against the ordinary column kernels, same process,
lab library (MIFFT_HERM / MIFFT_HS are re-read at every plan creation there).
    MIFFT_LIBRARY=hackathon_fft_amd/csrc/libmifft_lab.so python tools/herm_probe.py [shape ...]   (shape = 100x640x480)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hackathon_fft_amd as mf  # noqa: E402

SHAPES = ["25x640x480", "100x640x480", "200x640x480", "400x640x480", "10x1920x1080", "20x1920x1080", "40x1920x1080",
          "1x3840x2160", "4x3840x2160", "8x3840x2160", "50x64x64x64", "100x64x64x64", "200x64x64x64", "400x64x64x64",
          "10x128x128x128", "20x128x128x128", "40x128x128x128", "1x256x256x256", "4x256x256x256", "8x256x256x256",
          "1x512x512x512", "1000x96x80", "64x1024x1024", "16x2048x2048", "6x360x360x360"]


def main():
    shapes = [a for a in sys.argv[1:] if "x" in a] or SHAPES
    print(f"{'shape':>18} {'out MB':>8} {'off ms':>8} {'herm ms':>8} {'+hs ms':>8} {'all ms':>8} {'herm/':>7} {'+hs/':>7} {'all/off':>7}  kernels (all on)")
    for spec in shapes:
        shape = tuple(int(v) for v in spec.split("x"))
        x = torch.randn(shape + (1,), device="cuda:0")
        out = torch.empty(shape + (2,), device="cuda:0")
        res = {}
        # off / Hermitian last pass alone / with the half-store pass right before it / with the half-spectrum schedule of
        # three-pass plans as well (the default)
        for mode in ("0", "h", "a", "1", "0", "h", "a", "1"):
            os.environ["MIFFT_HERM"] = "0" if mode == "0" else os.environ.get("PROBE_HERM_ON", "1")  # (2: beyond the policy)
            os.environ["MIFFT_HS"] = "0" if mode in ("0", "h") else "1"
            os.environ["MIFFT_HERM_FIRST_AXIS"] = "1" if mode == "1" else "0"
            with mf.DeviceContext(0) as ctx:
                plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
                mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
                ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
                res[mode] = min(ms, res.get(mode, 1e9))
                name = [plan.kernel_name(d) for d in range(len(shape) - 2, -1, -1)]
            del plan
        print(f"{spec:>18} {out.numel() * 4 / 1e6:8.0f} {res['0']:8.4f} {res['h']:8.4f} {res['a']:8.4f} {res['1']:8.4f} "
              f"{res['h'] / res['0']:7.3f} {res['a'] / res['0']:7.3f} {res['1'] / res['0']:7.3f}  {name}", flush=True)
        del x, out


if __name__ == "__main__":
    main()
