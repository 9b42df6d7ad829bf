// kernels_generic.hip -- the "faithful stages" kernel family of libmifft (gfx950).
//
// Works for every factorisable length whose two LDS ping-pong rows fit the
// CU's 160 KiB, every radix (including large primes), fp32/fp64 output,
// real/complex and integer input, forward/inverse, contiguous and strided
// dimensions.  It executes the user's radix stages literally: one LDS pass per
// stage, one thread per output element, sequential complex-FMA accumulation in
// the order j = 1..R-1 -- the arithmetic of _radix_n_fft_kernel_stockham
// (fft/fft/_fft.mojo:228-296).  What is NOT the reference's design:
//   * a workgroup owns a TILE of transforms (several rows, or a block of
//     adjacent columns of a strided dimension) so that every HBM access is a
//     coalesced run and 64-lane waves stay full for small N (the reference uses
//     one N-thread block per row, fft/fft/_ndim_fft_gpu.mojo:549-564);
//   * strided dimensions are transformed in place inside LDS column tiles --
//     there is no transpose kernel and no scratch buffer
//     (reference: fft/fft/_ndim_fft_gpu.mojo:210-276, :185);
//   * twiddles come from a per-dimension device table computed in fp64 on the
//     host (the reference evaluates sin/cos in fp32 per use, _utils.mojo:85-99).
// Specialised register-butterfly kernels (kernels_fast_*.hip) take over the
// BASELINE shapes; this family is the general fallback and the
// MIFFT_FLAG_FAITHFUL_STAGES path.
#include "fft_radix.h"  // bf16_t
#include "mifft_internal.h"

namespace mifft {

template <typename T> struct C2;
template <> struct C2<float> { using type = float2; };
template <> struct C2<double> { using type = double2; };

struct GenericParams {
    const void* in;
    void* out;
    const void* tw;       // complex<T>[N]
    long long n_tiles;
    long long n_rows;     // ROWS: total rows;  COLS: unused
    long long inner;      // COLS: element stride of the dimension
    long long tiles_per_outer;  // COLS: ceil(inner / tile)
    int N;
    int tile;             // transforms per tile
    int ld;               // LDS leading dimension per transform (complex elements)
    int nstages;
    double scale;         // 1/N on the last stage of an inverse transform, else 1
    unsigned short radix[MIFFT_MAX_STAGES];
    unsigned int processed[MIFFT_MAX_STAGES];
};

template <typename T>
__device__ __forceinline__ T fma_t(T a, T b, T c) { return __builtin_fma(a, b, c); }
template <>
__device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// w * x + c as four real FMAs (Mojo ComplexSIMD.fma, call site fft/fft/_fft.mojo:290)
template <typename T, typename V>
__device__ __forceinline__ V cfma(V w, V x, V c) {
    V r;
    r.x = fma_t<T>(w.x, x.x, -fma_t<T>(w.y, x.y, -c.x));
    r.y = fma_t<T>(w.x, x.y, fma_t<T>(w.y, x.x, c.y));
    return r;
}

template <typename T, typename TIn>
__device__ __forceinline__ T cast_in(TIn v) {
    return (T)v;
}
template <typename T>
__device__ __forceinline__ T cast_in(bf16_t v) {
    return (T)(float)v;
}

// first-stage load with dtype cast and real->complex promotion (fft/fft/_fft.mojo:254-257)
template <typename T, typename TIn, int COMPS>
__device__ __forceinline__ typename C2<T>::type load_in(const void* base, long long g) {
    typename C2<T>::type v;
    const TIn* p = (const TIn*)base;
    // (float) first: exact for every element type narrower than T (and the only conversion bf16_t offers)
    if (COMPS == 1) {
        v.x = cast_in<T>(p[g]);
        v.y = (T)0;
    } else {
        v.x = cast_in<T>(p[2 * g]);
        v.y = cast_in<T>(p[2 * g + 1]);
    }
    return v;
}

// ROWS = contiguous dimension (tile = consecutive rows), else strided dimension
// (tile = `tile` adjacent columns of one outer block), transformed in place.
template <typename T, typename TIn, int COMPS, bool ROWS>
__global__ __launch_bounds__(256) void generic_kernel(const GenericParams p) {
    using V = typename C2<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V* bufA = (V*)smem;
    V* bufB = bufA + (size_t)p.tile * p.ld;
    const V* tw = (const V*)p.tw;
    const int N = p.N, ld = p.ld, tid = threadIdx.x, nthr = blockDim.x;

    for (long long t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
        long long base;
        int cn;
        if (ROWS) {
            long long r0 = t * p.tile;
            long long left = p.n_rows - r0;
            cn = (int)(left < p.tile ? left : p.tile);
            base = r0 * N;
        } else {
            long long o = t / p.tiles_per_outer;
            long long c0 = (t - o * p.tiles_per_outer) * p.tile;
            long long left = p.inner - c0;
            cn = (int)(left < p.tile ? left : p.tile);
            base = o * (long long)N * p.inner + c0;
        }
        const int total = cn * N;

        // ---- HBM -> LDS (coalesced runs) ----
        for (int f = tid; f < total; f += nthr) {
            int c, n;
            long long g;
            if (ROWS) {
                c = f / N;
                n = f - c * N;
                g = base + f;
            } else {
                n = f / cn;
                c = f - n * cn;
                g = base + (long long)n * p.inner + c;
            }
            bufA[c * ld + n] = load_in<T, TIn, COMPS>(p.in, g);
        }
        __syncthreads();

        // ---- the user's radix stages, literally ----
        V* src = bufA;
        V* dst = bufB;
        for (int s = 0; s < p.nstages; ++s) {
            const int R = p.radix[s], P = (int)p.processed[s];
            const int PR = P * R, step = N / R, ratio = N / PR;
            const bool last = s == p.nstages - 1;
            for (int f = tid; f < total; f += nthr) {
                const int c = f / N, i = f - c * N;
                const int q = i / PR, k = i - q * PR;   // k = i mod (P*R)
                const int n = q * P + k % P;            // fft/fft/_fft.mojo:233-235
                const V* sp = src + c * ld + n;
                V acc = sp[0];
                int jk = 0;
                for (int j = 1; j < R; ++j) {
                    jk += k;                             // j * (i mod PR)
                    if (jk >= PR) jk -= PR;              // ... mod PR   (:264-267)
                    acc = cfma<T, V>(tw[jk * ratio], sp[j * step], acc);
                }
                if (last) {                              // :292-294
                    acc.x *= (T)p.scale;
                    acc.y *= (T)p.scale;
                }
                dst[c * ld + i] = acc;
            }
            __syncthreads();
            V* tmp = src;
            src = dst;
            dst = tmp;
        }

        // ---- LDS -> HBM ----
        V* outp = (V*)p.out;
        for (int f = tid; f < total; f += nthr) {
            int c, n;
            long long g;
            if (ROWS) {
                c = f / N;
                n = f - c * N;
                g = base + f;
            } else {
                n = f / cn;
                c = f - n * cn;
                g = base + (long long)n * p.inner + c;
            }
            outp[g] = src[c * ld + n];
        }
        __syncthreads();
    }
}

// PREPARE = true: plan time, on the plan's device -- opt the instantiation in to > 64 KiB of dynamic LDS
// (per device) instead of launching, so that exec contains nothing but launches.
template <typename T, typename TIn, int COMPS, bool ROWS>
static hipError_t launch_one(const GenericParams& gp, int grid, int threads, size_t lds, hipStream_t stream,
                             bool prepare = false) {
    auto k = generic_kernel<T, TIn, COMPS, ROWS>;
    if (prepare) {
        if (lds > 64 * 1024)
            return hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return hipSuccess;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, stream, gp);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_first(const Plan& plan, const GenericParams& gp, int grid, int threads, size_t lds,
                               hipStream_t s, bool prep = false) {
    const int c = plan.in_components;
    switch (plan.in_dtype) {
        case MIFFT_F32:
            return c == 1 ? launch_one<T, float, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, float, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_F64:
            return c == 1 ? launch_one<T, double, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, double, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_U8:
            return c == 1 ? launch_one<T, unsigned char, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, unsigned char, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_I32:
            return c == 1 ? launch_one<T, int, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, int, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_I8:
            return c == 1 ? launch_one<T, signed char, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, signed char, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_I16:
            return c == 1 ? launch_one<T, short, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, short, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_U16:
            return c == 1 ? launch_one<T, unsigned short, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, unsigned short, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_F16:
            return c == 1 ? launch_one<T, _Float16, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, _Float16, 2, true>(gp, grid, threads, lds, s, prep);
        case MIFFT_BF16:
            return c == 1 ? launch_one<T, bf16_t, 1, true>(gp, grid, threads, lds, s, prep)
                          : launch_one<T, bf16_t, 2, true>(gp, grid, threads, lds, s, prep);
    }
    return hipErrorInvalidValue;
}

// dispatch shared by exec (prep = false) and plan-time preparation (prep = true)
static hipError_t dispatch_generic(const Plan& plan, const DimPass& pass, const GenericParams& gp, int grid,
                                   hipStream_t stream, bool prep) {
    const bool rows = pass.inner == 1;
    const bool f32 = plan.out_dtype == MIFFT_F32;
    if (pass.first)
        return f32 ? launch_first<float>(plan, gp, grid, pass.threads, pass.lds_bytes, stream, prep)
                   : launch_first<double>(plan, gp, grid, pass.threads, pass.lds_bytes, stream, prep);
    if (rows)
        return f32 ? launch_one<float, float, 2, true>(gp, grid, pass.threads, pass.lds_bytes, stream, prep)
                   : launch_one<double, double, 2, true>(gp, grid, pass.threads, pass.lds_bytes, stream, prep);
    return f32 ? launch_one<float, float, 2, false>(gp, grid, pass.threads, pass.lds_bytes, stream, prep)
               : launch_one<double, double, 2, false>(gp, grid, pass.threads, pass.lds_bytes, stream, prep);
}

static int launch_generic(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                          hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    GenericParams gp{};
    gp.in = in;
    gp.out = out;
    gp.tw = pass.d_twiddle;
    gp.N = (int)pass.N;
    gp.tile = pass.tile;
    gp.ld = pass.ld;
    gp.nstages = (int)pass.radices.size();
    gp.scale = plan.inverse ? 1.0 / (double)pass.N : 1.0;
    for (int s = 0; s < gp.nstages; ++s) {
        gp.radix[s] = (unsigned short)pass.radices[s];
        gp.processed[s] = pass.processed[s];
    }
    const bool rows = pass.inner == 1;
    if (rows) {
        gp.n_rows = count * pass.outer;
        gp.inner = 1;
        gp.tiles_per_outer = 1;
        gp.n_tiles = (gp.n_rows + pass.tile - 1) / pass.tile;
    } else {
        gp.n_rows = 0;
        gp.inner = pass.inner;
        gp.tiles_per_outer = (pass.inner + pass.tile - 1) / pass.tile;
        gp.n_tiles = count * pass.outer * gp.tiles_per_outer;
    }
    long long max_grid = (long long)plan.num_cus * 16;
    int grid = (int)(gp.n_tiles < max_grid ? gp.n_tiles : max_grid);
    hipError_t e = dispatch_generic(plan, pass, gp, grid, stream, /*prep=*/false);
    if (e != hipSuccess) return hip_error(e, "generic_kernel launch");
    return MIFFT_OK;
}

bool select_generic(const Plan& plan, DimPass& pass, std::string& why_not) {
    const size_t esz = plan.out_elem_bytes();
    const size_t max_lds = 160 * 1024;
    const bool rows = pass.inner == 1;
    const int64_t N = pass.N;
    if (pass.first && !rows) {
        why_not = "internal: first pass must be the contiguous dimension";
        return false;
    }
    // LDS row pitch: odd for column tiles so that the transposing HBM->LDS
    // writes of adjacent columns land on distinct banks.
    int64_t ld = rows ? N : (N | 1);
    if ((size_t)(2 * ld) * esz > max_lds) {
        why_not = "dimension of length " + std::to_string(N) + " does not fit two LDS rows (160 KiB)";
        return false;
    }
    const int64_t budget = 64 * 1024;  // two buffers; keeps >= 2 workgroups per CU
    int64_t tile = budget / (int64_t)(2 * ld * (int64_t)esz);
    if (tile < 1) tile = 1;
    if (rows) {
        if (tile > 64) tile = 64;
    } else {
        if (tile > 16) tile = 16;
        if (tile > pass.inner) tile = pass.inner;
    }
    pass.tile = (int)tile;
    pass.ld = (int)ld;
    pass.threads = 256;
    pass.lds_bytes = (size_t)(2 * tile * ld) * esz;
    pass.kernel_name = "generic";
    pass.launch = launch_generic;
    if (pass.lds_bytes > 64 * 1024) {  // plan creation runs on the plan's device
        GenericParams none{};
        hipError_t e = dispatch_generic(plan, pass, none, 1, nullptr, /*prep=*/true);
        if (e != hipSuccess) {
            why_not = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e);
            return false;
        }
    }
    return true;
}

}  // namespace mifft
