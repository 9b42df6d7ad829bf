"""Runtime-specialised row kernels (default: lengths with a prime radix 37 ... 4093): time per 128 MB for a set of lengths
(PROBE_LENGTHS=a,b,c overrides the list), once per value of a kernel-header
macro (lab build: MIFFT_JIT_DEFINES, one process per value so that the runtime-compiled kernels differ).
    python tools/big_prime_probe2.py [-DMIFFT_BIGP_SB=4 -DMIFFT_BIGP_SB=8 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, hackathon_fft_amd as mf
def factors(n):
    f, d = [], 2
    while n > 1:
        if n %% d == 0:
            f.append(d)
            while n %% d == 0: n //= d
        d += 1
    return f
row = []
lengths = [int(v) for v in os.environ.get("PROBE_LENGTHS", "").split(",") if v] or \
    [37, 61, 83, 97, 101, 113, 127, 131, 194, 262, 251, 509, 521, 1009, 2018, 1517, 4093]
for n in lengths:
    batch = max(1, int(128e6 / (n * 8)))
    x = torch.randn((batch, n, 2), device="cuda:0"); out = torch.empty_like(x)
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=[factors(n)], ctx=ctx)
        mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
        got = torch.view_as_complex(out[:4].contiguous()).cpu().numpy()
        ref = np.fft.fft(torch.view_as_complex(x[:4].contiguous()).cpu().numpy().astype(np.complex128), axis=1)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        ms = mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
    row.append("%%d:%%.4f%%s" %% (n, ms, "" if err < 1e-5 else "(ERR %%.1e)" %% err))
print(os.environ.get("MIFFT_JIT_DEFINES", "") + " rader_min=" + os.environ.get("MIFFT_RADER_MIN", "default"), " ".join(row), flush=True)
''' % ROOT
for d in (sys.argv[1:] or [""]):   # "-D..." = header macro for the runtime compiler, "NAME=value" = environment switch
    env = dict(os.environ, MIFFT_LIBRARY=os.path.join(ROOT, "hackathon_fft_amd", "csrc", "libmifft_lab.so"))
    if d.startswith("-D"):
        env["MIFFT_JIT_DEFINES"] = d
    elif "=" in d:
        env[d.split("=")[0]] = d.split("=", 1)[1]
    subprocess.run([sys.executable, "-c", CHILD], env=env)
