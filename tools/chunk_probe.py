"""Does running an N-D batch in chunks (all passes of a few images back to back, so that the later passes find
`out` in L2 / Infinity Cache) beat whole-batch passes?  GPU box only:  python tools/chunk_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hackathon_fft_amd as mf

def run(shape, chunks_list, reps=30):
    x = torch.randn(shape + (2,), device="cuda:0")
    out = torch.empty_like(x)
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
        b = shape[0]
        for nch in chunks_list:
            if nch > b:
                continue
            bounds = [(i * b // nch, (i + 1) * b // nch) for i in range(nch)]
            for _ in range(3):
                for lo, hi in bounds:
                    mf.fft(out, x, ctx, plan=plan, first=lo, count=hi - lo)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for lo, hi in bounds:
                    mf.fft(out, x, ctx, plan=plan, first=lo, count=hi - lo)
            e1.record()
            torch.cuda.synchronize()
            print(f"{shape} chunks {nch:3d}: {e0.elapsed_time(e1) / reps:.4f} ms per batch")

run((100, 640, 480), [1, 2, 4, 5, 10, 20])
run((10, 128, 128, 128), [1, 2, 5, 10])
run((100, 64, 64, 64), [1, 2, 4, 10, 25])
run((10, 1920, 1080), [1, 2, 5, 10])
