"""Randomised end-to-end check of the GPU path against fp64 pocketfft (GPU box):
    python tools/fuzz_gpu.py [cases] [seed] [big]
Random ranks 1-5, arbitrary lengths (so every kernel family is hit: tables, runtime-specialised rows / column tiles /
planes, literal stages, four-step), fp32 / fp64, real / complex / uint8 / int32 input, forward / inverse, ragged batches.
Prints every failure and a per-family count; exit status 1 on any failure."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hackathon_fft_amd as mf

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# "big": 2-D ... 4-D shapes of 2-30 M points, mostly real input -- large enough for the plan-time policy to take the Hermitian
# last pass and the half-store pass in front of it (herm_pays, mifft_internal.h)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
rng = np.random.default_rng(seed)
fam = collections.Counter()
fails = 0
t_start = time.time()
for i in range(cases):
    if i and i % 100 == 0:  # (a run that prints nothing for minutes is taken to be hung)
        print(f"... {i} cases, {fails} failures, {time.time() - t_start:.0f} s", flush=True)
    nd = int(rng.choice([1, 1, 1, 2, 2, 2, 3, 3, 4, 5]))
    if BIG:
        nd = int(rng.choice([2, 2, 3, 3, 4]))
        pool = [32, 48, 50, 64, 64, 80, 96, 100, 120, 128, 128, 160, 200, 243, 256, 256, 320, 360, 384, 480, 500, 512, 640, 720, 1000, 1024, 1080]
        cap = {2: 1100, 3: 260, 4: 70}[nd]
        shape = tuple(int(rng.choice([v for v in pool if v <= cap])) for _ in range(nd))
        pts = int(np.prod(shape))
        batch = int(max(1, min(400, rng.integers(2_000_000, 30_000_000) // pts)))
    elif nd == 1:
        n = int(rng.choice([rng.integers(2, 700), rng.integers(2, 5000), 2 ** int(rng.integers(1, 15)), rng.integers(16385, 70000)]))
        shape = (n,)
        batch = int(rng.integers(1, max(2, min(300, 200000 // n))))
    elif nd == 2:
        shape = (int(rng.integers(2, 300)), int(rng.integers(2, 300)))
        batch = int(rng.integers(1, 6))
    elif nd == 3:
        shape = tuple(int(rng.integers(2, 50)) for _ in range(3))
        batch = int(rng.integers(1, 4))
    else:  # rank 4 / 5 like the reference's commented-out bench shapes (fft/bench.mojo:120-121)
        shape = tuple(int(rng.integers(2, 20 if nd == 4 else 12)) for _ in range(nd))
        batch = int(rng.integers(1, 3))
    out_dt = np.float32 if rng.random() < 0.65 else np.float64
    kind = rng.choice(["c", "c", "r", "u8", "i32", "mixed"])
    if BIG:
        kind = rng.choice(["r", "r", "r", "u8", "i32", "c"])
    inverse = bool(rng.random() < 0.3)
    comps = 2
    if kind == "c":
        x = rng.standard_normal((batch,) + shape + (2,)).astype(out_dt)
    elif kind == "r":
        comps = 1
        x = rng.standard_normal((batch,) + shape + (1,)).astype(out_dt)
    elif kind == "u8":
        comps = int(rng.choice([1, 2]))
        x = rng.integers(0, 255, size=(batch,) + shape + (comps,)).astype(np.uint8)
    elif kind == "i32":
        comps = int(rng.choice([1, 2]))
        x = rng.integers(-1000, 1000, size=(batch,) + shape + (comps,)).astype(np.int32)
    else:
        x = rng.standard_normal((batch,) + shape + (2,)).astype(np.float32 if out_dt == np.float64 else np.float64)
    xd = torch.from_numpy(x).to("cuda:0")
    odt = torch.float32 if out_dt == np.float32 else torch.float64
    out = torch.full((batch,) + shape + (2,), float("nan"), device="cuda:0", dtype=odt)
    try:
        plan = mf.plan_fft(xd.dtype, odt, xd.shape, out.shape, inverse=inverse)
        mf.fft(out, xd, plan=plan)
        torch.cuda.synchronize()
    except mf.MifftError as e:
        # -9: documented size limits (a prime factor too large for the LDS rows of the literal stages, ...);
        # -5 / -7: the reference's own behaviour -- its default "gpu" radix estimate (trial division by 2..32,
        # fft/fft/fft.mojo:61-80) does not factor lengths with a larger prime factor; the user must pass bases
        cat = "limit" if e.status == -9 else "default-bases" if e.status in (-5, -7) else "error"
        if cat != "default-bases":
            print(f"{cat} {shape} batch {batch} {kind} {out_dt.__name__} inv={inverse}: {e}")
        fails += 1 if cat == "error" else 0
        fam[cat] += 1
        continue
    names = [plan.kernel_name(d) for d in range(nd)]
    if "generic" in names and os.environ.get("FUZZ_SHOW_GENERIC"):
        print(f"generic: {shape} batch {batch} {kind} {out_dt.__name__} {names}")
    for nm in set(names):
        if nm.endswith(("_h", "_h_jit")):
            fam["hermitian"] += 1
        if "_hs" in nm:
            fam["half-store"] += 1
        key = "generic" if nm == "generic" else "jit" if nm.endswith("_jit") else "transpose" if nm == "transpose" else "table"
        if "_ts" in nm:
            key = "fourstep-" + key
        if nm.startswith("plane"):
            key = "plane-" + key
        fam[key] += 1
    xc = x[..., 0].astype(np.float64) + (1j * x[..., 1].astype(np.float64) if comps == 2 else 0)
    axes = tuple(range(1, nd + 1))
    truth = np.fft.ifftn(xc, axes=axes) if inverse else np.fft.fftn(xc, axes=axes)
    got = out.cpu().numpy().astype(np.float64)
    gc = got[..., 0] + 1j * got[..., 1]
    num = np.linalg.norm((gc - truth).reshape(batch, -1), axis=1)
    den = np.linalg.norm(truth.reshape(batch, -1), axis=1) + 1e-300
    err = float((num / den).max())
    tol = 1e-5 if out_dt == np.float32 else 1e-11
    if not np.isfinite(got).all() or err > tol:
        fails += 1
        print(f"FAIL {shape} batch {batch} {kind} {out_dt.__name__} inv={inverse} {names}: rel l2 {err:.3e}")
print(f"{cases} cases, {fails} failures, {time.time() - t_start:.0f} s; kernel families hit: {dict(fam)}")
sys.exit(1 if fails else 0)
