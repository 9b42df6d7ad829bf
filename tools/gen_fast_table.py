#!/usr/bin/env python3
"""Generate hackathon_fft_amd/csrc/fast_table_gen_{rows,cols}.inc: fused tile-kernel configurations
for the common smooth lengths that have no hand-tuned entry in kernels_fast.hip.

Heuristics (from the MI355X tuning runs recorded in DESIGN.md):
  * 2-4 composite passes, every radix <= 16 (register budget), as few passes as possible for rows,
    small radices for column tiles (one big-LDS workgroup per CU cannot hide register stalls);
  * rows: 256 threads, tile ~ 4096 elements (32 KiB LDS) -> 16 elements per thread;
  * column tiles: 16 adjacent columns (128-B HBM runs), thread count for <= ~20 elements/thread;
  * twiddles in the compact LDS table; first/last pass go straight to HBM when lanes cover >= 16
    consecutive elements (128-B runs), otherwise through a flat coalesced LDS copy.
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "hackathon_fft_amd", "csrc")

HAND_ROWS = {1024, 512, 256, 128, 64, 2048, 4096, 93, 480, 640}
HAND_COLS = {640, 480, 128, 64, 256, 1024}

SIZES = sorted(set(
    [2 ** k for k in range(4, 15)] +
    [24, 40, 48, 60, 72, 80, 96, 100, 120, 144, 160, 192, 200, 240, 250, 320, 360, 384, 400, 500, 600, 720, 768, 800,
     900, 960, 1000, 1080, 1200, 1280, 1440, 1536, 1600, 1920, 2000, 2160, 2400, 2560, 2880, 3000, 3072, 3200, 3840,
     4000, 4320, 5120, 6144, 7680]))

ALLOWED = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16]


def factorizations(n, k, allowed):
    """ordered-insensitive multisets of k factors from `allowed` with product n"""
    res = []

    def rec(rem, start, acc):
        if len(acc) == k:
            if rem == 1:
                res.append(tuple(acc))
            return
        for i in range(start, len(allowed)):
            r = allowed[i]
            if rem % r == 0:
                rec(rem // r, i, acc + [r])
    rec(n, 0, [])
    return res


def choose(n, cols, allowed=None):
    best = None
    for k in (2, 3, 4):
        for f in factorizations(n, k, allowed or ALLOWED):
            mx = max(f)
            if cols:
                # small radices first priority; 4 passes fine
                score = (mx > 10, mx > 8, k if mx <= 8 else 9, mx, -min(f))
            else:
                score = (mx > 16, k, mx, -min(f))
            if best is None or score < best[0]:
                best = (score, f)
    if best is None:
        return None
    f = sorted(best[1], reverse=not cols)
    return f


def pow2ceil(v):
    p = 1
    while p < v:
        p *= 2
    return p


ALLOWED_F64 = [2, 3, 4, 5, 6, 7, 8, 9, 10]   # 16-byte elements: keep butterflies small


def row_cfg(n, f64=False):
    f = choose(n, False, ALLOWED_F64 if f64 else None)
    if f is None:
        return None
    budget = 2048 if f64 else 4096          # elements per tile: 32 KiB of LDS either way
    if n <= budget:
        tile = max(1, min(64, budget // n))
        threads = 256
        # keep every pass busy: items of the largest-radix pass >= threads/2
        while tile * (n // max(f)) < threads // 2 and tile < 64:
            tile *= 2
    else:
        if n > (4096 if f64 else 8192):
            return None  # 16 elements/thread needs 1024 threads = 128 VGPRs: spills
        tile = 1
        threads = min(512, pow2ceil(n // (8 if f64 else 16)))
        if not f64 and n >= 6144:
            # one 48-64-KiB row + an equally large twiddle table per workgroup = one workgroup per CU: 1024 threads
            # and the next row prefetched into registers (8192 points: 0.137 -> 0.122 ms, tools/tune GROUP 14)
            fd = (n // f[0]) >= 16
            ld = (n // f[-1]) >= 16
            return f, tile, 1024, fd, ld, True
    # pass 0 reads HBM directly even when a butterfly's lanes cover only a few consecutive elements: N = 93 (runs of
    # 3 elements) gained 15 % over the flat LDS staging copy (tools/tune GROUP 2)
    fd = os.environ.get("MIFFT_GEN_FD", "1") == "1" or (n // f[0]) >= 16
    ld = (n // f[-1]) >= 16
    return f, tile, threads, fd, ld


def twl_entries(f):
    """entries of the compact LDS twiddle table: sum over passes k >= 1 of P_k * (R_k - 1)"""
    tot, p = 0, 1
    for k, r in enumerate(f):
        if k >= 1:
            tot += p * (r - 1)
        p *= r
    return tot


def choose3(n, allowed):
    """3 passes, radices <= 16, ascending (column tiles with 32 elements per thread spill with 4 passes)"""
    best = None
    for f in factorizations(n, 3, [a for a in (allowed or ALLOWED) if a <= 16]):
        score = (max(f), -min(f))
        if best is None or score < best[0]:
            best = (score, f)
    return sorted(best[1]) if best else None


def col_cfg(n, f64=False):
    f = choose(n, True, ALLOWED_F64 if f64 else None)
    if f is None:
        return None
    big = n * (8 if f64 else 16) > (4096 if f64 else 8192)   # would need > 16 (8) elements per thread at 16 columns
    if big:
        f3 = choose3(n, ALLOWED_F64 if f64 else None)
        if f3 is not None:
            f = f3
    tile = 8 if f64 else 16                 # 128-byte HBM runs
    esz = 16 if f64 else 8
    # <= 512 threads; up to 32 (16 for f64) elements per thread for 3-pass configurations (4-pass ones spill)
    cap = (8192 if f64 else 16384) if len(f) <= 3 else (4096 if f64 else 8192)
    while (n * tile > cap or n * tile * esz > 136 * 1024) and tile > 2:
        tile //= 2
    if n * tile > cap:
        return None
    per = 10 if f64 else 20
    threads = max(128, min(512, pow2ceil((n * tile + per - 1) // per)))
    if n * tile * esz > 136 * 1024:
        return None
    return f, tile, threads


def wide_col_cfgs(n, f64=False):
    """Short strided dimensions also get WIDE tiles (32 / 64 columns: 256- / 512-byte HBM runs) for strides that are a
    multiple of the tile width (select_fast takes a wide entry only then): 128-point tiles on the z axis of 10 x 128^3
    0.0597 -> 0.0537 ms (tools/tune GROUP 18).  At most 4096 elements (32 KiB) per tile, two passes preferred."""
    base = col_cfg(n, f64)
    if base is None:
        return []
    res = []
    cap = 2048 if f64 else 4096
    for w in ((32, 16) if f64 else (64, 32)):
        if n * w > cap:
            continue
        f = None
        for k in (2, 3):
            cand = [c for c in factorizations(n, k, ALLOWED_F64 if f64 else ALLOWED) if max(c) <= (8 if f64 else 16)]
            if cand:
                f = sorted(min(cand, key=lambda c: (max(c), -min(c))), reverse=True)
                break
        if f is None:
            f = base[0]
        per = 8 if f64 else 16
        threads = max(128, min(512, pow2ceil((n * w + per - 1) // per)))
        res.append((f, w, threads))
    return res


def emit(name, n, f, tile, threads, cols, fd, ld, f64=False, pf=False, tag="", macro="MIFFT_CFG"):
    r = list(f) + [1] * (4 - len(f))
    rs = "x".join(str(v) for v in f) + tag
    ty, dt, suffix = ("double", "MIFFT_F64", "_f64") if f64 else ("float", "MIFFT_F32", "")
    esz = 16 if f64 else 8
    # the compact twiddle table must fit next to the tile (160 KiB per workgroup); else read the global table
    twm = "TW_LDS" if (n * tile + twl_entries(f)) * esz <= 156 * 1024 else "TW_GLOBAL"
    return (f'    {macro}("{name}{n}{suffix}_{rs}", {ty}, {dt}, {n}, {len(f)}, {r[0]}, {r[1]}, {r[2]}, {r[3]}, '
            f'{tile}, {threads}, {"true" if cols else "false"}, {"true" if fd else "false"}, '
            f'{"true" if ld else "false"}, {twm}, 1, {"true" if pf else "false"}),')


HAND_ROWS_F64 = {1024, 512, 256, 128, 64, 93, 480, 640}
HAND_COLS_F64 = {640, 480, 128, 64, 256}
SIZES_F64 = [n for n in SIZES if n <= 4096]


def main():
    rows, cols, rows64, cols64 = [], [], [], []
    for n in SIZES:
        if n not in HAND_ROWS:
            c = row_cfg(n)
            if c:
                # non-temporal-store twin for the 0.25-0.55 GB window of batched 1-D transforms (fast_table.h), listed first
                # (not when the last pass stores short runs that are no whole number of 128-byte lines: kernels_jit.cpp)
                run = (n // c[0][-1]) * 8
                if (not c[4]) or run % 128 == 0 or run >= 512:
                    rows.append(emit("rows", n, c[0], c[1], c[2], False, c[3], c[4], pf=len(c) > 5 and c[5], macro="MIFFT_CFG_MID_ST"))
                rows.append(emit("rows", n, c[0], c[1], c[2], False, c[3], c[4], pf=len(c) > 5 and c[5]))
        if n not in HAND_COLS and n <= 8192:
            for w in wide_col_cfgs(n):   # wide tiles first: select_fast takes the first entry that fits the stride
                cols.append(emit("cols", n, w[0], w[1], w[2], True, True, True, tag=f"_w{w[1]}"))
            c = col_cfg(n)
            if c:
                cols.append(emit("cols", n, c[0], c[1], c[2], True, True, True))
    for n in SIZES_F64:
        if n not in HAND_ROWS_F64:
            c = row_cfg(n, True)
            if c:
                run = (n // c[0][-1]) * 16
                if (not c[4]) or run % 128 == 0 or run >= 512:
                    rows64.append(emit("rows", n, c[0], c[1], c[2], False, c[3], c[4], True, macro="MIFFT_CFG_MID_ST"))
                rows64.append(emit("rows", n, c[0], c[1], c[2], False, c[3], c[4], True))
        if n not in HAND_COLS_F64 and n <= 2048:
            for w in wide_col_cfgs(n, True):
                cols64.append(emit("cols", n, w[0], w[1], w[2], True, True, True, True, tag=f"_w{w[1]}"))
            c = col_cfg(n, True)
            if c:
                cols64.append(emit("cols", n, c[0], c[1], c[2], True, True, True, True))
    hdr = "// GENERATED by tools/gen_fast_table.py -- do not edit.\n"
    open(os.path.join(OUT, "fast_table_gen_rows.inc"), "w").write(hdr + "\n".join(rows) + "\n")
    open(os.path.join(OUT, "fast_table_gen_cols.inc"), "w").write(hdr + "\n".join(cols) + "\n")
    open(os.path.join(OUT, "fast_table_gen_rows_f64.inc"), "w").write(hdr + "\n".join(rows64) + "\n")
    open(os.path.join(OUT, "fast_table_gen_cols_f64.inc"), "w").write(hdr + "\n".join(cols64) + "\n")
    print(len(rows), "row configs,", len(cols), "column configs,", len(rows64), "f64 rows,", len(cols64), "f64 cols")
    for l in rows[:60]:
        print(l.strip()[:110])


if __name__ == "__main__":
    main()
