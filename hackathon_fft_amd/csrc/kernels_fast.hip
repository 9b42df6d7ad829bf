// kernels_fast.hip -- register-butterfly Stockham tile kernels for MI355X (gfx950).
//
// One kernel per dimension.  A workgroup owns a TILE of independent transforms of one
// dimension; the tile lives in LDS for the whole transform and HBM is touched exactly once
// for reading and once for writing:
//
//   pass 0      : work item = one radix-R0 butterfly; its R0 inputs come straight from HBM
//                 (lanes -> consecutive elements, so every wave instruction reads whole
//                 contiguous runs), DFT_R0 in registers, results scattered into LDS at the
//                 Stockham-permuted positions        dst[q*P*R + s*P + p]
//   pass 1..k-2 : LDS -> registers (stride N/R gather, conflict-free), twiddle W_{P R}^{j p},
//                 DFT_R in registers, registers -> LDS (same buffer, after a barrier)
//   pass k-1    : as above, but the results go straight to HBM in natural order
//                 (for the last stage q = 0, so lanes again write contiguous runs).
//
// The user's radix stages (reference: one LDS pass + barrier per stage, one thread per
// output, fft/fft/_ndim_fft_gpu.mojo:359-386) are fused into 2-4 composite passes, e.g.
// 1024 = 2^10 -> 16 * 8 * 8: two LDS exchanges instead of ten.  Work items of a pass are
// flattened over the whole workgroup, so radices that do not divide the thread count
// (31 * 3, 10 * 6 * 8) keep every lane busy.
//
// Strided dimensions (COLS): the tile is TILE adjacent columns, LDS layout [n][column];
// lanes run along the columns, so HBM runs are TILE*8 bytes and LDS accesses are
// contiguous by construction.  The transform is in place -- this replaces the reference's
// transpose kernel + scratch buffer (fft/fft/_ndim_fft_gpu.mojo:210-276, :185).
//
// Twiddles: W_N^n from the plan's fp64-accurate table; per-thread twiddles are loop
// invariant over tiles, so a persistent workgroup loads them into registers once.
// Inverse: conj(F(conj x)) * 1/N with the forward butterflies (bit-identical to conjugated
// twiddles), reference semantics fft/fft/_utils.mojo:101-104, fft/fft/_fft.mojo:292-294.
#include "fft_radix.h"
#include "mifft_internal.h"

namespace mifft {

struct TileParams {
    const void* in;
    void* out;
    const void* tw;  // cpx<T>[N], conjugated when the plan is an inverse plan
    long long n_tiles;
    long long n_rows;           // ROWS
    long long inner;            // COLS
    long long tiles_per_outer;  // COLS
    int inverse;
    double scale;  // 1/N for inverse
};

constexpr int ilog2_ce(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}
constexpr bool is_pow2_ce(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T_, int N_, int NP_, int R0_, int R1_, int R2_, int R3_, int TILE_, int THREADS_, bool COLS_,
          bool FIRST_DIRECT_, bool LAST_DIRECT_, bool TWREG_, int ROWPAD_ = 0>
struct TileCfg {
    using T = T_;
    static constexpr int N = N_, NP = NP_, TILE = TILE_, THREADS = THREADS_;
    static constexpr bool COLS = COLS_, FIRST_DIRECT = FIRST_DIRECT_, LAST_DIRECT = LAST_DIRECT_, TWREG = TWREG_;
    static constexpr int LD = N_ + ROWPAD_;  // ROWS: LDS pitch of one transform
    static constexpr int R(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : i == 2 ? R2_ : R3_; }
    static constexpr int P(int i) {
        int p = 1;
        for (int k = 0; k < i; ++k) p *= R(k);
        return p;
    }
    static constexpr int NB(int i) { return N / R(i); }
    static constexpr int ITEMS(int i) { return NB(i) * TILE; }
    static constexpr int IPT(int i) { return (ITEMS(i) + THREADS - 1) / THREADS; }
    static constexpr int TW_OFF(int i) {  // register twiddles of passes 1..i-1 precede pass i
        int o = 0;
        for (int k = 1; k < i; ++k) o += IPT(k) * (R(k) - 1);
        return o;
    }
    static constexpr int TW_TOTAL = TW_OFF(NP_);
    static constexpr size_t LDS_BYTES = (size_t)(COLS_ ? N_ * TILE_ : LD * TILE_) * 2 * sizeof(T_);
    static_assert(P(NP_) == N_, "radices must multiply to N");
};

// XOR swizzle of the in-row index for the exchange written by pass E (power-of-two rows
// only): the 16 lanes of a ds_write_b64 group own 16 butterflies whose outputs are P*R
// elements apart; fold the low butterfly bits into the bank-selecting low 4 index bits.
template <class C, int E>
MIFFT_DEV int swz(int n) {
    if constexpr (!C::COLS && is_pow2_ce(C::N) && E >= 0 && E < C::NP - 1) {
        constexpr int a = ilog2_ce(C::P(E)), c = ilog2_ce(C::P(E) * C::R(E));
        constexpr int c4 = c < 4 ? c : 4, hi = c > 4 ? c : 4, nb = c4 - a;
        if constexpr (nb > 0 && (1 << hi) < C::N) {
            return n ^ (((n >> hi) & ((1 << nb) - 1)) << a);
        } else {
            return n;
        }
    } else {
        return n;
    }
}

template <class C, int E>
MIFFT_DEV int lds_index(int c, int n) {
    if constexpr (C::COLS)
        return n * C::TILE + c;
    else
        return c * C::LD + swz<C, E>(n);
}

template <class C>
MIFFT_DEV long long gaddr(const TileParams& p, long long base, int c, int n) {
    if constexpr (C::COLS)
        return base + (long long)n * p.inner + c;
    else
        return base + (long long)c * C::N + n;
}

template <class C, int I>
MIFFT_DEV void item_decode(int id, int& c, int& b) {
    if constexpr (C::COLS) {
        b = id / C::TILE;
        c = id - b * C::TILE;
    } else {
        c = id / C::NB(I);
        b = id - c * C::NB(I);
    }
}

template <class C, int I>
MIFFT_DEV void preload_tw(cpx<typename C::T>* twr, const cpx<typename C::T>* tw, int tid, int inverse) {
    if constexpr (I < C::NP) {
        constexpr int R = C::R(I), P = C::P(I), RATIO = C::N / (P * R);
#pragma unroll
        for (int k = 0; k < C::IPT(I); ++k) {
            int id = tid + k * C::THREADS, c, b;
            if (id >= C::ITEMS(I)) id = 0;
            item_decode<C, I>(id, c, b);
            const int pp = b % P;
#pragma unroll
            for (int j = 1; j < R; ++j) {
                cpx<typename C::T> w = tw[j * pp * RATIO];
                if (inverse) w.y = -w.y;  // plan table is conjugated for inverse plans; we need W forward
                twr[C::TW_OFF(I) + k * (R - 1) + (j - 1)] = w;
            }
        }
        preload_tw<C, I + 1>(twr, tw, tid, inverse);
    }
}

template <class C, int I>
MIFFT_DEV void run_pass(const TileParams& p, cpx<typename C::T>* lds, const cpx<typename C::T>* twr, long long base,
                        int nv, int tid) {
    if constexpr (I < C::NP) {
        using T = typename C::T;
        using V = cpx<T>;
        constexpr int R = C::R(I), P = C::P(I), NB = C::NB(I), IPT = C::IPT(I), RATIO = C::N / (P * R);
        constexpr bool EXACT = C::ITEMS(I) % C::THREADS == 0;
        constexpr bool SRC_GLOBAL = (I == 0) && C::FIRST_DIRECT;
        constexpr bool DST_GLOBAL = (I == C::NP - 1) && C::LAST_DIRECT;
        V v[IPT][R];
        const V* gin = (const V*)p.in;
        V* gout = (V*)p.out;

        // ---- gather the R inputs of every butterfly this thread owns ----
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int id = tid + k * C::THREADS;
            if (EXACT || id < C::ITEMS(I)) {
                int c, b;
                item_decode<C, I>(id, c, b);
                if constexpr (SRC_GLOBAL) {
                    const bool ok = c < nv;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        V x = {(T)0, (T)0};
                        if (ok) x = gin[gaddr<C>(p, base, c, b + j * NB)];
                        if (p.inverse) x.y = -x.y;
                        v[k][j] = x;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < R; ++j) v[k][j] = lds[lds_index<C, I - 1>(c, b + j * NB)];
                }
                if constexpr (I > 0) {
                    const int pp = b % P;
#pragma unroll
                    for (int j = 1; j < R; ++j) {
                        V w;
                        if constexpr (C::TWREG) {
                            w = twr[C::TW_OFF(I) + k * (R - 1) + (j - 1)];
                        } else {
                            w = ((const V*)p.tw)[j * pp * RATIO];
                            if (p.inverse) w.y = -w.y;
                        }
                        v[k][j] = cmul(v[k][j], w);
                    }
                }
            }
        }
        // in-place LDS buffer: every read of this pass completes before any later write (this pass's
        // scatter, or pass 0 of the NEXT tile when this pass stores to HBM)
        if constexpr (!SRC_GLOBAL) __syncthreads();

        // ---- butterflies + Stockham scatter ----
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int id = tid + k * C::THREADS;
            if (EXACT || id < C::ITEMS(I)) {
                int c, b;
                item_decode<C, I>(id, c, b);
                Dft<R, T, 1>::run(v[k]);
                const int q = b / P, pp = b - q * P;
                const int o0 = q * P * R + pp;
                if constexpr (DST_GLOBAL) {
                    if (c < nv) {
#pragma unroll
                        for (int s = 0; s < R; ++s) {
                            V y = v[k][s];
                            if (p.inverse) {
                                y.x *= (T)p.scale;
                                y.y *= -(T)p.scale;
                            }
                            gout[gaddr<C>(p, base, c, o0 + s * P)] = y;
                        }
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < R; ++s) lds[lds_index<C, I>(c, o0 + s * P)] = v[k][s];
                }
            }
        }
        if constexpr (!DST_GLOBAL) __syncthreads();
        run_pass<C, I + 1>(p, lds, twr, base, nv, tid);
    }
}

template <class C>
__global__ __launch_bounds__(C::THREADS) void tile_kernel(const TileParams p) {
    using T = typename C::T;
    using V = cpx<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V* lds = (V*)smem;
    const int tid = threadIdx.x;
    V twr[C::TW_TOTAL > 0 ? C::TW_TOTAL : 1];
    if constexpr (C::TWREG) preload_tw<C, 1>(twr, (const V*)p.tw, tid, p.inverse);

    for (long long t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
        long long base;
        int nv;
        if constexpr (C::COLS) {
            const long long o = t / p.tiles_per_outer;
            const long long c0 = (t - o * p.tiles_per_outer) * C::TILE;
            const long long left = p.inner - c0;
            nv = (int)(left < C::TILE ? left : C::TILE);
            base = o * (long long)C::N * p.inner + c0;
        } else {
            const long long r0 = t * C::TILE;
            const long long left = p.n_rows - r0;
            nv = (int)(left < C::TILE ? left : C::TILE);
            base = r0 * C::N;
        }
        if constexpr (!C::FIRST_DIRECT) {
            // flat, fully coalesced HBM -> LDS copy of the tile (rows need not be 16-B aligned: N = 93)
            static_assert(!C::COLS || C::FIRST_DIRECT, "column tiles always load directly");
            const V* gin = (const V*)p.in;
            const int total = nv * C::N;
            for (int f = tid; f < total; f += C::THREADS) {
                const int c = f / C::N, n = f - c * C::N;
                V x = gin[base + f];
                if (p.inverse) x.y = -x.y;
                lds[lds_index<C, -1>(c, n)] = x;
            }
            __syncthreads();
        }
        run_pass<C, 0>(p, lds, twr, base, nv, tid);
        if constexpr (!C::LAST_DIRECT) {
            static_assert(!C::COLS || C::LAST_DIRECT, "column tiles always store directly");
            V* gout = (V*)p.out;
            const int total = nv * C::N;
            for (int f = tid; f < total; f += C::THREADS) {
                const int c = f / C::N, n = f - c * C::N;
                V y = lds[lds_index<C, C::NP - 1>(c, n)];
                if (p.inverse) {
                    y.x *= (T)p.scale;
                    y.y *= -(T)p.scale;
                }
                gout[base + f] = y;
            }
            __syncthreads();
        }
    }
}

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
    if (C::COLS) {
        tp.inner = pass.inner;
        tp.tiles_per_outer = (pass.inner + C::TILE - 1) / C::TILE;
        tp.n_tiles = count * pass.outer * tp.tiles_per_outer;
    } else {
        tp.n_rows = count * pass.outer;
        tp.inner = 1;
        tp.tiles_per_outer = 1;
        tp.n_tiles = (tp.n_rows + C::TILE - 1) / C::TILE;
    }
    auto k = tile_kernel<C>;
    static bool attr_set = false;
    if (C::LDS_BYTES > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return hip_error(e, "hipFuncSetAttribute");
        attr_set = true;
    }
    // persistent workgroups: enough to fill every CU to its LDS / wave limit
    long long per_cu = (160 * 1024) / (long long)(C::LDS_BYTES ? C::LDS_BYTES : 1);
    const long long wave_limit = 2048 / C::THREADS;
    if (per_cu > wave_limit) per_cu = wave_limit;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)plan.num_cus * per_cu;
    if (grid > tp.n_tiles) grid = tp.n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::THREADS), C::LDS_BYTES, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "tile_kernel launch");
    return MIFFT_OK;
}

// ---------------------------------------------------------------------------------------------
// configuration table.  NP, R0..R3, TILE, THREADS, COLS, FIRST_DIRECT, LAST_DIRECT, TWREG
// ---------------------------------------------------------------------------------------------
struct FastEntry {
    int out_dtype;
    int N;
    bool cols;
    const char* name;
    LaunchFn launch;
    int tile, threads;
    size_t lds;
};

#define MIFFT_CFG(NAME, T, DT, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWREG)                         \
    {                                                                                                               \
        DT, N, COLS, NAME, launch_tile<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWREG>>,   \
            TILE, THREADS, TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THREADS, COLS, FD, LD_, TWREG>::LDS_BYTES       \
    }

static const FastEntry kFastTable[] = {
    // ---- contiguous dimension, fp32 ----
    MIFFT_CFG("rows1024_16x8x8", float, MIFFT_F32, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, true),
    MIFFT_CFG("rows512_8x8x8", float, MIFFT_F32, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, true),
    MIFFT_CFG("rows256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, false, true, true, true),
    MIFFT_CFG("rows128_8x4x4", float, MIFFT_F32, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, true),
    MIFFT_CFG("rows64_4x4x4", float, MIFFT_F32, 64, 3, 4, 4, 4, 1, 32, 256, false, true, true, true),
    MIFFT_CFG("rows2048_16x16x8", float, MIFFT_F32, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, true),
    MIFFT_CFG("rows4096_16x16x16", float, MIFFT_F32, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, true),
    MIFFT_CFG("rows93_31x3", float, MIFFT_F32, 93, 2, 31, 3, 1, 1, 64, 192, false, false, false, false),
    MIFFT_CFG("rows480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 128, false, true, true, false),
    MIFFT_CFG("rows640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, false, true, true, false),
    // ---- strided dimensions, fp32 ----
    MIFFT_CFG("cols640_10x8x8", float, MIFFT_F32, 640, 3, 10, 8, 8, 1, 8, 256, true, true, true, false),
    MIFFT_CFG("cols480_10x6x8", float, MIFFT_F32, 480, 3, 10, 6, 8, 1, 8, 256, true, true, true, false),
    MIFFT_CFG("cols128_16x8", float, MIFFT_F32, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, true),
    MIFFT_CFG("cols64_8x8", float, MIFFT_F32, 64, 2, 8, 8, 1, 1, 16, 128, true, true, true, true),
    MIFFT_CFG("cols256_16x16", float, MIFFT_F32, 256, 2, 16, 16, 1, 1, 16, 256, true, true, true, true),
};

bool select_fast(const Plan& plan, DimPass& pass) {
    // fast families take complex input of the output dtype; everything else (real / integer
    // input, mixed precision) runs on the generic family
    if (pass.first && (plan.in_components != 2 || plan.in_dtype != plan.out_dtype)) return false;
    const bool cols = pass.inner != 1;
    for (const FastEntry& e : kFastTable) {
        if (e.out_dtype != plan.out_dtype || e.N != pass.N || e.cols != cols) continue;
        if (cols && pass.inner < e.tile) continue;
        pass.kernel_name = e.name;
        pass.launch = e.launch;
        pass.tile = e.tile;
        pass.threads = e.threads;
        pass.lds_bytes = e.lds;
        pass.ld = (int)pass.N;
        return true;
    }
    return false;
}

}  // namespace mifft
