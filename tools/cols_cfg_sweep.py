"""Sweep runtime-specialised COLUMN configurations (radices, tile, threads) of one strided length (lab build:
MIFFT_JIT_COLS_CFG + MIFFT_SKIP_GEN_TABLE), complex 2-D shape B x N x D2, whole transform and the table's own choice beside it.
    MIFFT_LIBRARY=hackathon_fft_amd/csrc/libmifft_lab.so python tools/cols_cfg_sweep.py 10x1920x1080 [max_configs]"""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hackathon_fft_amd as mf  # noqa: E402


def multisets(n, k, cap, start=2):
    if k == 1:
        return [[n]] if start <= n <= cap else []
    res = []
    for r in range(start, cap + 1):
        if n % r == 0:
            res += [[r] + t for t in multisets(n // r, k - 1, cap, r)]
    return res


def timed(x, out):
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=5, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx) for _ in range(3))
        return ms, plan.kernel_name(0)


def main():
    spec = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    shape = tuple(int(v) for v in spec.split("x"))
    n = shape[1]
    x = torch.randn(shape + (2,), device="cuda:0")
    out = torch.empty(shape + (2,), device="cuda:0")
    os.environ.pop("MIFFT_JIT_COLS_CFG", None)
    os.environ["MIFFT_SKIP_GEN_TABLE"] = "0"
    base, name = timed(x, out)
    print(f"{spec}: table {base:.4f} ms {name}", flush=True)
    os.environ["MIFFT_SKIP_GEN_TABLE"] = "1"
    cands = []
    for k, cap in ((3, 16), (4, 10), (2, 32)):
        for ms_ in multisets(n, k, cap):
            for order in (sorted(ms_), sorted(ms_, reverse=True)):
                if order not in [c[0] for c in cands]:
                    cands.append((order, k))
    configs = []
    for f, k in cands:
        for tile in (16, 8, 4):
            if n * tile * 8 > 136 * 1024 or n * tile > (16384 if k <= 3 else 8192) * 2:
                continue
            for threads in (512, 256, 1024):
                per = n * tile / threads / max(f)  # butterflies of the widest pass per thread
                if per < 0.5 or n * tile / threads > 40:
                    continue
                configs.append((f, tile, threads))
    configs = configs[:limit]
    results = []
    for f, tile, threads in configs:
        cfg = "x".join(str(v) for v in f) + f":{tile}:{threads}"
        os.environ["MIFFT_JIT_COLS_CFG"] = cfg
        try:
            ms, name = timed(x, out)
        except Exception as e:  # a configuration the kernel rejects (LDS, static asserts)
            print(f"   {cfg:>22}  failed: {str(e)[:80]}", flush=True)
            continue
        results.append((ms, cfg, name))
        print(f"   {cfg:>22} {ms:8.4f} ms  {ms / base:6.3f}  {name}", flush=True)
    results.sort()
    print("best:", [(f"{ms:.4f}", cfg) for ms, cfg, _ in results[:5]], flush=True)


if __name__ == "__main__":
    main()
