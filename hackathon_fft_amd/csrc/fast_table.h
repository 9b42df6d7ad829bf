// fast_table.h -- launcher + table-entry plumbing shared by the translation units that
// instantiate tile_kernel<> configurations (kernels_fast.hip and the generated tables).
#pragma once

#include "mifft_internal.h"
#include "tile_kernel.h"

namespace mifft {

template <class C>
static int launch_tile(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                       hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    TileParams tp{};
    tp.in = in;
    tp.out = out;
    tp.tw = pass.d_twiddle;
    tp.inverse = plan.inverse;
    tp.scale = plan.inverse ? 1.0 / (double)pass.N : 1.0;
    tp.tlo = pass.d_aux;   // TSTORE configurations only
    tp.thi = pass.d_aux2;
    tp.tcol = pass.d_aux3;
    tp.reverse = pass.reverse;
    tp.store_lim = pass.store_lim;  // (HS configurations)
    if (C::COLS) {
        tp.inner = pass.inner;
        tp.tiles_per_outer = (pass.inner + C::TILE - 1) / C::TILE;
        if (C::HERM) {  // last pass of a real-input N-D plan: only the columns up to their mirror are transformed
            tp.herm_d0 = pass.herm_d0;
            tp.herm_d1 = pass.herm_d1;
            tp.herm_d2 = pass.herm_d2;
            tp.herm_dj = pass.herm_dj;
            tp.herm_js = pass.herm_js;
            tp.herm_L = pass.herm_L;
            tp.herm_H = pass.herm_H;
            tp.herm_tpr = (pass.herm_H + C::TILE - 1) / C::TILE;
            tp.tiles_per_outer = herm_tiles_per_outer(pass, C::TILE);
        } else if (pass.col_prefix > 0) {  // (middle pass of a half-spectrum schedule: only the columns that were stored)
            tp.tiles_per_outer = (pass.col_prefix + C::TILE - 1) / C::TILE;
            tp.col_lim = pass.col_prefix;
        }
        tp.n_tiles = count * pass.outer * tp.tiles_per_outer;
    } else {
        tp.n_rows = count * pass.outer;
        tp.inner = 1;
        tp.tiles_per_outer = 1;
        tp.n_tiles = (tp.n_rows + C::TILE - 1) / C::TILE;
    }
    auto k = tile_kernel<C>;
    // workgroups per CU of the persistent grid: the LDS / wave-count formula unless the table carries a measured value for
    // this kernel (kGridPerCu, kernels_fast.hip)
    const long long grid = tile_grid<C>(plan.num_cus, tp.n_tiles, pass.wg_per_cu);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::THREADS), C::LDS_BYTES, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "tile_kernel launch");
    return MIFFT_OK;
}

// ---------------------------------------------------------------------------------------------
// configuration table.  NP, R0..R3, TILE, THREADS, COLS, FIRST_DIRECT, LAST_DIRECT, TWMODE, MINW, PREFETCH
// ---------------------------------------------------------------------------------------------
// plan-time preparation on the plan's (current) device: kernels whose tile needs more than 64 KiB of dynamic LDS
// must opt in per device.  Done here, not in exec, so that exec contains nothing but launches (graph capture).
template <class C>
static int prepare_tile() {
    if (C::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)tile_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::LDS_BYTES);
        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return MIFFT_OK;
}

struct FastEntry {
    bool tstore;      // four-step first pass: transposed + twiddled store (needs pass.d_aux / d_aux2)
    bool in_real;     // the kernel promotes a real (C_in = 1) tensor in its pass-0 load
    int stream_pref;  // 1: only for streaming-size problems (non-temporal twin), 2: only as the first pass of a
                      // cache-resident N-D transform (non-temporal LOADS), 3 / 4: batched 1-D transforms that move
                      // 0.25-0.55 GB / more than 0.05 GB per exec (non-temporal STORES), -1: any size
    int out_dtype;
    int N;
    bool cols;
    const char* name;
    LaunchFn launch;
    int (*prepare)();
    int tile, threads;
    size_t lds;
    bool herm = false;  // TileCfg::HERM twin: last (strided, in-place) pass of a real-input 2-D .. 4-D plan
    bool hs = false;    // TileCfg::HS twin: the pass before it, storing only the lower half of its dimension
};

#define MIFFT_TILECFG(TS, REAL, NTM, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF) \
    TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, REAL, false, NTM, TS>
#define MIFFT_CFG_X(TS, REAL, NTM, STREAM, NAME, T, DT, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF) \
    {                                                                                                                      \
        TS, REAL, STREAM, DT, N, COLS, NAME,                                                                               \
            launch_tile<MIFFT_TILECFG(TS, REAL, NTM, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)>,  \
            prepare_tile<MIFFT_TILECFG(TS, REAL, NTM, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)>, \
            TILE, THREADS,                                                                                                 \
            MIFFT_TILECFG(TS, REAL, NTM, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)::LDS_BYTES     \
    }

#define MIFFT_CFG(...) MIFFT_CFG_X(false, false, 0, -1, __VA_ARGS__)
// Hermitian twin of a strided configuration (TileCfg::HERM), optionally with wave-owned sub-problems
#define MIFFT_CFG_HERM_X(WS, NAME, T, DT, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)                              \
    {                                                                                                                                      \
        false, false, -1, DT, N, COLS, NAME "_h",                                                                                          \
            launch_tile<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, WS, false, 0, true>>,  \
            prepare_tile<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, WS, false, 0, true>>, \
            TILE, THREADS,                                                                                                                 \
            TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, WS, false, 0, true>::LDS_BYTES,     \
            true                                                                                                                           \
    }
// half-store twin (TileCfg::HS) of a row / column configuration: REAL = promotes a real tensor, NTM = non-temporal mode,
// STREAM = stream_pref of the entry
#define MIFFT_HS_TILECFG(REAL, NTM, WS, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF) \
    TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, REAL, false, NTM, false, T, WS, false, 0, false, true>
#define MIFFT_CFG_HS_X(REAL, NTM, STREAM, WS, NAME, T, DT, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)                \
    {                                                                                                                                      \
        false, REAL, STREAM, DT, N, COLS, NAME "_hs",                                                                                      \
            launch_tile<MIFFT_HS_TILECFG(REAL, NTM, WS, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)>,            \
            prepare_tile<MIFFT_HS_TILECFG(REAL, NTM, WS, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)>,           \
            TILE, THREADS,                                                                                                                 \
            MIFFT_HS_TILECFG(REAL, NTM, WS, T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)::LDS_BYTES, false, true   \
    }
#define MIFFT_CFG_HS(...) MIFFT_CFG_HS_X(false, 0, -1, false, __VA_ARGS__)
#define MIFFT_CFG_HERM(...) MIFFT_CFG_HERM_X(false, __VA_ARGS__)
#define MIFFT_CFG_WSUB_HERM(...) MIFFT_CFG_HERM_X(true, __VA_ARGS__)
// column tile whose passes 1..NP-1 run inside wave-owned sub-problems (TileCfg::WSUB): R0 a multiple of THREADS / 64
#define MIFFT_CFG_WSUB(NAME, T, DT, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF)                        \
    {                                                                                                                          \
        false, false, -1, DT, N, COLS, NAME,                                                                                   \
            launch_tile<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, true>>,   \
            prepare_tile<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, true>>,  \
            TILE, THREADS,                                                                                                     \
            TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWM, MINW, PF, 0, false, false, 0, false, T, true>::LDS_BYTES       \
    }
// complex-input kernel + its real-input twin (contiguous dimension only)
#define MIFFT_CFG_CR(NAME, ...) \
    MIFFT_CFG_X(false, false, 0, -1, NAME, __VA_ARGS__), MIFFT_CFG_X(false, true, 0, -1, NAME "_r", __VA_ARGS__)
// non-temporal twin for problems that dwarf the Infinity Cache (listed BEFORE the plain entry)
#define MIFFT_CFG_STREAM(NAME, ...) MIFFT_CFG_X(false, false, 3, 1, NAME "_nt", __VA_ARGS__)
// ... for real input (the reference's own benchmark is the real-input one, fft/bench.mojo:57-97): 0.2407 -> 0.2326 ms at
// 100k x 1024, against 0.2095 ms for a promoting copy of the same bytes (tools/micro/rw_mix.hip)
#define MIFFT_CFG_STREAM_R(NAME, ...) MIFFT_CFG_X(false, true, 3, 1, NAME "_r_nt", __VA_ARGS__)
#define MIFFT_CFG_STREAM_ST_R(NAME, ...) MIFFT_CFG_X(false, true, 2, 1, NAME "_r_nts", __VA_ARGS__)
#define MIFFT_CFG_NTL_R(NAME, ...) MIFFT_CFG_X(false, true, 1, 2, NAME "_r_ntl", __VA_ARGS__)
// ... non-temporal STORES for batched 1-D transforms in the window where plain stores thrash the 256-MB Infinity Cache
// (stream_pref 3: ONE dimension, 0.25 ... 0.55 GB moved per exec): the output is not read again by this plan, and keeping
// it out of the cache leaves the cache to the loads.  30k x 1024 0.0935 -> 0.0756 ms, 250k x 128 0.0893 -> 0.0778,
// generated lengths 49 ... 3125 at 0.4 GB 6-20 % faster; neutral or slower below 0.25 GB and (for the generated
// configurations) above 0.55 GB (tools/tune GROUPs 1-3 with TUNE_BATCH, tools/nts_probe.py).  Non-temporal LOADS lose below
// ~0.6 GB and stay with the streaming twins above.  N-D plans never take these: their later passes read `out` back.
#define MIFFT_CFG_MID_ST(NAME, ...) MIFFT_CFG_X(false, false, 2, 3, NAME "_nts", __VA_ARGS__)
#define MIFFT_CFG_MID_ST_R(NAME, ...) MIFFT_CFG_X(false, true, 2, 3, NAME "_r_nts", __VA_ARGS__)
// ... the same from 0.05 GB on (stream_pref 4): N = 93, whose flat-copy store gains at every size measured
#define MIFFT_CFG_SMALL_ST(NAME, ...) MIFFT_CFG_X(false, false, 2, 4, NAME "_nts", __VA_ARGS__)
#define MIFFT_CFG_SMALL_ST_R(NAME, ...) MIFFT_CFG_X(false, true, 2, 4, NAME "_r_nts", __VA_ARGS__)
// ... with non-temporal stores only (tiles staged through a flat LDS copy re-read their lines)
#define MIFFT_CFG_STREAM_ST(NAME, ...) MIFFT_CFG_X(false, false, 2, 1, NAME "_nts", __VA_ARGS__)
// ... with non-temporal loads only: first pass of an N-D transform whose `out` fits the Infinity Cache
#define MIFFT_CFG_NTL(NAME, ...) MIFFT_CFG_X(false, false, 1, 2, NAME "_ntl", __VA_ARGS__)
// column tile with the transposed + twiddled store (first pass of the four-step); pass LAST_DIRECT = false
#define MIFFT_CFG_TS(NAME, ...) MIFFT_CFG_X(true, false, 0, -1, NAME "_ts", __VA_ARGS__)

// tables instantiated in kernels_fast_gen_rows.hip / kernels_fast_gen_cols.hip (host-only data:
// kept TU-local there so that the device pass never sees the host launcher pointers)
const FastEntry* gen_rows_table(int* count);
const FastEntry* gen_cols_table(int* count);
const FastEntry* gen_rows_f64_table(int* count);
const FastEntry* gen_cols_f64_table(int* count);
// column tile of length pass.N with the transposed + twiddled store, or false
bool select_fast_tstore(const Plan& plan, DimPass& pass);

}  // namespace mifft
