"""Packed real rows (TileCfg::R2C): sweep radices / tile / threads / prefetch of the N / 2-point row tile through
MIFFT_JIT_ROWS_CFG (lab build), real-input shape B x D0 x N, whole transform; the tuned half-store kernel (MIFFT_R2C=0) beside it.
    MIFFT_LIBRARY=hackathon_fft_amd/csrc/libmifft_lab.so python tools/r2c_rows_sweep.py 100x640x480 [max_configs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hackathon_fft_amd as mf  # noqa: E402
from tools.cols_cfg_sweep import multisets  # noqa: E402


def timed(x, out):
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=5, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx) for _ in range(3))
        return ms, plan.kernel_name(len(x.shape) - 3)


def main():
    spec = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    shape = tuple(int(v) for v in spec.split("x"))
    n = shape[-1] // 2
    x = torch.randn(shape + (1,), device="cuda:0")
    out = torch.empty(shape + (2,), device="cuda:0")
    os.environ.pop("MIFFT_JIT_ROWS_CFG", None)
    os.environ["MIFFT_R2C"] = "0"
    base, name = timed(x, out)
    print(f"{spec}: without packed rows {base:.4f} ms {name}", flush=True)
    os.environ["MIFFT_R2C"] = "1"
    ms, name = timed(x, out)
    print(f"{spec}: default packed rows {ms:.4f} ms {name}", flush=True)
    cands = []
    for k, cap in ((2, 16), (3, 16), (3, 10), (4, 8)):
        for ms_ in multisets(n, k, cap):
            order = sorted(ms_, reverse=True)
            if order not in cands:
                cands.append(order)
    configs = []
    for f in cands:
        for tile, threads in ((8, 128), (8, 256), (16, 256), (16, 512), (32, 512), (4, 128)):
            per = n * tile / threads
            if per < 4 or per > 32 or n * tile * 8 > 64 * 1024:
                continue
            for pf in (0, 1):
                configs.append((f, tile, threads, pf))
    results = []
    for f, tile, threads, pf in configs[:limit]:
        cfg = "x".join(str(v) for v in f) + f":{tile}:{threads}:{pf}"
        os.environ["MIFFT_JIT_ROWS_CFG"] = cfg
        try:
            ms, name = timed(x, out)
        except Exception as e:
            print(f"   {cfg:>22}  failed: {str(e)[:80]}", flush=True)
            continue
        results.append((ms, cfg))
        print(f"   {cfg:>22} {ms:8.4f} ms  {ms / base:6.3f}  {name}", flush=True)
    results.sort()
    print("best:", [(f"{ms:.4f}", cfg) for ms, cfg in results[:5]], flush=True)


if __name__ == "__main__":
    main()
