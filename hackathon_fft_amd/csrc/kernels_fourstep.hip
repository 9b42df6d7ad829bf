// kernels_fourstep.hip -- a contiguous dimension longer than one workgroup's LDS row (> 16 384 points).
//
// N = N1 * N2, x viewed as [N1][N2] (n = n1*N2 + n2):
//   pass 1  column FFTs of length N1 (stride N2)            x       -> scratch   A[k1][n2]
//   pass 2  transpose + twiddle  C[n2][k1] = A[k1][n2] * W_N^{k1*n2}  scratch -> out
//   pass 3  column FFTs of length N2 (stride N1), in place   out               X[k1 + N1*k2] at [k2][k1]
// which leaves the spectrum in natural order with no trailing transpose; both FFT passes are the in-place
// column-tile kernels that already serve the strided dimensions of N-D transforms.  The reference has no
// equivalent off NVIDIA thread-block clusters (fft/fft/_ndim_fft_gpu.mojo:100-108, 510-519); this is
// SURVEY.md 8(f) item 3.  W_N^m comes from a two-level table W_N^{m mod L} * W_N^{L*(m div L)}, L = 1024,
// both factors rounded once from long double (k1*n2 < N, so no modular reduction is needed).
#include <cmath>
#include <cstdlib>

#include "fast_table.h"
#include "mifft_config.h"

namespace mifft {

static constexpr int kL = 1024;

// The plan scratch (one tensor of the output size).  In the lab build (-DMIFFT_TESTING) MIFFT_TEST_FAIL_SCRATCH_ALLOC=1 makes
// this allocation fail the way an exhausted device does, so that the roll-back of a half-built route can be tested on any GPU.
static hipError_t alloc_scratch(Plan& plan) {
    if (plan.d_scratch) return hipSuccess;
    plan.scratch_bytes = (size_t)plan.batch * (size_t)plan.prod * plan.out_elem_bytes();
    if (!plan.scratch_bytes) return hipSuccess;
#ifdef MIFFT_TESTING  // fault injection exists in the lab build only
    if (config().test_fail_scratch_alloc) {
        plan.scratch_bytes = 0;
        return hipErrorOutOfMemory;
    }
#endif
    hipError_t e = hipMalloc(&plan.d_scratch, plan.scratch_bytes);
    if (e != hipSuccess) {
        plan.d_scratch = nullptr;
        plan.scratch_bytes = 0;
        (void)hipGetLastError();
    }
    return e;
}

static void free_pass_tables(DimPass& q) {
    if (q.d_twiddle) (void)hipFree(q.d_twiddle);
    if (q.d_aux) (void)hipFree(q.d_aux);
    if (q.d_aux2) (void)hipFree(q.d_aux2);
    if (q.d_aux3) (void)hipFree(q.d_aux3);
    q.d_twiddle = q.d_aux = q.d_aux2 = q.d_aux3 = nullptr;
}

struct TTParams {
    const void* src;
    void* dst;
    const void* tlo;  // W_N^l, l in [0, L)
    const void* thi;  // W_N^{L*h}, h in [0, ceil(N/L))
    long long n1, n2, batch;
    int inverse;
    int apply_tw;  // 0: plain batched transpose (long strided dimensions), 1: four-step twiddle W_N^{k1*n2}
};

template <typename T>
__global__ __launch_bounds__(256) void transpose_twiddle_kernel(const TTParams p) {
    using V = cpx<T>;
    __shared__ V tile[32][33];
    const V* src = (const V*)p.src;
    V* dst = (V*)p.dst;
    const V* tlo = (const V*)p.tlo;
    const V* thi = (const V*)p.thi;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long long c0 = (long long)blockIdx.x * 32;
    const long long row_blocks = (p.n1 + 31) / 32;
    // grid.y is capped at 65535 blocks: a workgroup walks the row blocks y, y + gridDim.y, ...
    for (long long rb = blockIdx.y; rb < row_blocks; rb += gridDim.y)
    for (long long b = blockIdx.z; b < p.batch; b += gridDim.z) {
        const long long r0 = rb * 32;
        const V* s = src + b * p.n1 * p.n2;
        V* d = dst + b * p.n1 * p.n2;
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const long long k1 = r0 + ty + i, n2 = c0 + tx;
            if (k1 < p.n1 && n2 < p.n2) {
                V v = s[k1 * p.n2 + n2];
                if (p.apply_tw) {
                    const long long m = k1 * n2;  // < N
                    V wl = tlo[m & (kL - 1)], wh = thi[m >> 10];
                    V w = cmul(wl, wh);
                    if (p.inverse) w.y = -w.y;
                    v = cmul(v, w);
                }
                tile[ty + i][tx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const long long n2 = c0 + ty + i, k1 = r0 + tx;
            if (k1 < p.n1 && n2 < p.n2) d[n2 * p.n1 + k1] = tile[tx][ty + i];
        }
        __syncthreads();
    }
}

static int launch_transpose_twiddle(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                                    hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    TTParams tp{};
    tp.src = in;
    tp.dst = out;
    tp.tlo = pass.d_aux;
    tp.thi = pass.d_aux2;
    tp.n1 = pass.fs_n1;
    tp.n2 = pass.fs_n2;
    tp.batch = count * pass.outer;  // matrices per exec
    tp.inverse = plan.inverse;
    tp.apply_tw = pass.d_aux != nullptr;
    const long long row_blocks = (pass.fs_n1 + 31) / 32;
    dim3 grid((unsigned)((pass.fs_n2 + 31) / 32), (unsigned)(row_blocks < 65535 ? row_blocks : 65535),
              (unsigned)(tp.batch < 4096 ? tp.batch : 4096));
    if (plan.out_dtype == MIFFT_F32)
        hipLaunchKernelGGL(transpose_twiddle_kernel<float>, grid, dim3(256), 0, stream, tp);
    else
        hipLaunchKernelGGL(transpose_twiddle_kernel<double>, grid, dim3(256), 0, stream, tp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "transpose_twiddle launch");
    return MIFFT_OK;
}

template <typename T>
static hipError_t upload_two_level(int64_t N, void** d_lo, void** d_hi) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    const int64_t nhi = (N + kL - 1) / kL;
    std::vector<T> lo(2 * kL), hi(2 * (size_t)nhi);
    for (int64_t l = 0; l < kL; ++l) {
        long double th = -two_pi * (long double)l / (long double)N;
        lo[2 * l] = (T)cosl(th);
        lo[2 * l + 1] = (T)sinl(th);
    }
    for (int64_t h = 0; h < nhi; ++h) {
        long double th = -two_pi * (long double)(h * kL) / (long double)N;
        hi[2 * h] = (T)cosl(th);
        hi[2 * h + 1] = (T)sinl(th);
    }
    hipError_t e = hipMalloc(d_lo, lo.size() * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(*d_lo, lo.data(), lo.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(d_hi, hi.size() * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(*d_hi, hi.data(), hi.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// [tile][n1] table of W_N^(c * k1): the twiddle of column c0 + c of a TSTORE tile relative to column c0
template <typename T>
static hipError_t upload_col_table(int64_t N, int64_t n1, int tile, void** d_tab) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<T> tab(2 * (size_t)tile * (size_t)n1);
    for (int c = 0; c < tile; ++c)
        for (int64_t k1 = 0; k1 < n1; ++k1) {
            long double th = -two_pi * (long double)((int64_t)c * k1) / (long double)N;
            tab[2 * ((size_t)c * n1 + k1)] = (T)cosl(th);
            tab[2 * ((size_t)c * n1 + k1) + 1] = (T)sinl(th);
        }
    hipError_t e = hipMalloc(d_tab, tab.size() * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(*d_tab, tab.data(), tab.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// one column-tile pass of length n with element stride `inner`: fused table first, literal-stage fallback
static bool make_cols_pass(const Plan& plan, int dim_index, int64_t n, int64_t inner, DimPass& ps, std::string& why,
                           bool allow_jit = false) {
    ps = DimPass();
    ps.dim_index = dim_index;
    ps.N = n;
    ps.inner = inner;
    ps.outer = 1;
    for (int k = 0; k < dim_index; ++k) ps.outer *= plan.dims[k];  // 1 for a batched 1-D transform
    ps.first = false;
    if (select_fast(plan, ps)) return true;
    if (allow_jit) {
        std::string whyj;
        if (select_jit(plan, ps, whyj)) return true;
    }
    std::vector<uint64_t> user = plan_estimate_bases((uint64_t)n, true);
    std::string err;
    if (plan_ordered_bases((uint64_t)n, user, ps.radices, ps.processed, err) != MIFFT_OK) {
        why = "factor " + std::to_string(n) + ": " + err;
        return false;
    }
    return select_generic(plan, ps, why);
}

bool build_fourstep(Plan& plan, int dim_index, std::string& why_not) {
    const int64_t N = plan.dims[dim_index];
    // real / integer / mixed-precision input is widened by the first pass, which then has to be the runtime-specialised
    // transposed-store kernel (two-launch form); the three-launch fallback reads complex input of the output dtype
    const bool native_in = plan.in_components == 2 && plan.in_dtype == plan.out_dtype;
    // N = N1 * N2, both factors at most 4096, as balanced as possible, preferring lengths with fused kernels
    int64_t best1 = 0, best2 = 0;
    double best_score = 1e300;
    for (int64_t n2 = 2; n2 <= 4096 && n2 < N; ++n2) {
        if (N % n2) continue;
        const int64_t n1 = N / n2;
        if (n1 > 4096 || n1 < 2) continue;
        DimPass a, b;
        std::string w;
        Plan probe = plan;  // selection only reads the plan
        probe.passes.clear();
        const bool fa = (make_cols_pass(probe, dim_index, n1, n2, a, w) && std::string(a.kernel_name) != "generic") ||
                        jit_cols_feasible(plan, n1, n2);
        const bool fb = (make_cols_pass(probe, dim_index, n2, n1, b, w) && std::string(b.kernel_name) != "generic") ||
                        jit_cols_feasible(plan, n2, n1);
        DimPass ts;
        ts.N = n1;
        ts.inner = n2;
        // no middle pass, no scratch
        const bool two_pass = fb && ((native_in && select_fast_tstore(probe, ts)) || jit_tstore_feasible(plan, n1, n2));
        if (!native_in && !two_pass) continue;
        double score = std::fabs(std::log((double)n1 / (double)n2)) + (fa || two_pass ? 0 : 4) + (fb ? 0 : 4) -
                       (two_pass ? 3 : 0);
        if (score < best_score) {
            best_score = score;
            best1 = n1;
            best2 = n2;
        }
    }
    if (!best1) {
        why_not = native_in ? "no factorisation N1 * N2 with both factors <= 4096"
                            : "no two-pass factorisation N1 * N2 (needed for real / integer / mixed-precision input)";
        return false;
    }
    DimPass p1, p3, p2;
    if (!make_cols_pass(plan, dim_index, best2, best1, p3, why_not, true)) return false;
    // two-pass form: the first column pass stores transposed + twiddled straight into `out` (no scratch)
    {
        DimPass ts;
        ts.dim_index = dim_index;
        ts.N = best1;
        ts.inner = best2;
        ts.outer = p3.outer;
        ts.first = !native_in;  // reads x with its own element type / component count
        std::string whyts;
        if ((native_in && select_fast_tstore(plan, ts)) || select_jit_tstore(plan, ts, whyts)) {
            ts.src_buf = 0;  // x
            ts.dst_buf = 1;  // out
            p3.src_buf = 1;
            p3.dst_buf = 1;
            if (ts.prepare && ts.prepare() != MIFFT_OK) return false;
            if (p3.prepare && p3.prepare() != MIFFT_OK) return false;
            const bool inv2 = plan.inverse != 0;
            hipError_t e2 = upload_twiddle_table(plan.out_dtype, ts.N, inv2, &ts.d_twiddle);
            if (e2 == hipSuccess) e2 = upload_twiddle_table(plan.out_dtype, p3.N, inv2, &p3.d_twiddle);
            if (e2 == hipSuccess)
                e2 = plan.out_dtype == MIFFT_F32 ? upload_two_level<float>(N, &ts.d_aux, &ts.d_aux2)
                                                 : upload_two_level<double>(N, &ts.d_aux, &ts.d_aux2);
            if (e2 == hipSuccess)
                e2 = plan.out_dtype == MIFFT_F32 ? upload_col_table<float>(N, ts.N, ts.tile, &ts.d_aux3)
                                                 : upload_col_table<double>(N, ts.N, ts.tile, &ts.d_aux3);
            if (e2 != hipSuccess) {
                free_pass_tables(ts);
                free_pass_tables(p3);
                why_not = std::string("device allocation: ") + hipGetErrorString(e2);
                plan.alloc_failed = true;
                return false;
            }
            plan.passes.push_back(ts);
            plan.passes.push_back(p3);
            return true;
        }
    }
    if (!native_in) {
        why_not = "the transposed-store kernel could not be built for real / integer / mixed-precision input";
        return false;
    }
    if (!make_cols_pass(plan, dim_index, best1, best2, p1, why_not, true)) return false;
    p1.src_buf = 0;  // x
    p1.dst_buf = 2;  // scratch
    p3.src_buf = 1;  // out, in place
    p3.dst_buf = 1;
    p2.dim_index = dim_index;
    p2.outer = p3.outer;
    p2.N = N;
    p2.fs_n1 = best1;
    p2.fs_n2 = best2;
    p2.kernel_name = "transpose_twiddle";
    p2.launch = launch_transpose_twiddle;
    p2.src_buf = 2;
    p2.dst_buf = 1;

    const bool inv = plan.inverse != 0;
    hipError_t e = hipSuccess;
    if (p1.prepare && p1.prepare() != MIFFT_OK) return false;
    if (p3.prepare && p3.prepare() != MIFFT_OK) return false;
    e = upload_twiddle_table(plan.out_dtype, p1.N, inv, &p1.d_twiddle);
    if (e == hipSuccess) e = upload_twiddle_table(plan.out_dtype, p3.N, inv, &p3.d_twiddle);
    if (e == hipSuccess)
        e = plan.out_dtype == MIFFT_F32 ? upload_two_level<float>(N, &p2.d_aux, &p2.d_aux2)
                                        : upload_two_level<double>(N, &p2.d_aux, &p2.d_aux2);
    if (e == hipSuccess) e = alloc_scratch(plan);
    if (e != hipSuccess) {
        free_pass_tables(p1);
        free_pass_tables(p2);
        free_pass_tables(p3);
        why_not = std::string("device allocation: ") + hipGetErrorString(e);
        plan.alloc_failed = true;
        return false;
    }
    plan.passes.push_back(p1);
    plan.passes.push_back(p2);
    plan.passes.push_back(p3);
    return true;
}


// ---------------------------------------------------------------------------------------------
// Four-step for a STRIDED dimension too long for one column tile (N > 4096: the columns of 8K frames, ...).
// N = N1 * N2; the dimension is viewed as [N1][N2] rows of `inner` contiguous elements:
//   pass A  column tiles of N1 points (row stride N2 * inner), out -> scratch, row k1 of the tile (n2, columns) stored
//           as row n2 * N1 + k1 and multiplied by W_N^(k1 n2)             (TileCfg::FS1, runtime-specialised)
//   pass B  column tiles of N2 points (row stride N1 * inner), scratch -> out: row k2 * N1 + k1 = X[k1 + N1 k2]
// Two passes with TILE-element runs on both sides instead of the three of the transposed route below (transpose, row
// kernel, transpose) -- SURVEY.md 8(f).3; the reference has no path at all for such dimensions off NVIDIA clusters
// (fft/fft/_ndim_fft_gpu.mojo:100-108, 510-519).
// ---------------------------------------------------------------------------------------------
bool build_fourstep_strided(Plan& plan, int dim_index, std::string& why_not) {
    const int64_t N = plan.dims[dim_index];
    int64_t inner = 1, outer = 1;
    for (int k = dim_index + 1; k < plan.ndim; ++k) inner *= plan.dims[k];
    for (int k = 0; k < dim_index; ++k) outer *= plan.dims[k];
    if (inner == 1) {
        why_not = "contiguous dimension";
        return false;
    }
    if (!config().fourstep_strided) {  // (lab switch: forces the transposed route)
        why_not = "MIFFT_FOURSTEP_STRIDED=0";
        return false;
    }
    // most balanced factorisation whose factors both have a fused column configuration (Config::fs_n1 forces the first
    // factor: lab knob)
    int64_t best1 = 0, best2 = 0;
    const int64_t forced1 = config().fs_n1;
    double best_score = 1e300;
    for (int64_t n1 = 2; n1 <= 4096 && n1 < N; ++n1) {
        if (N % n1 || (forced1 > 0 && n1 != forced1)) continue;
        const int64_t n2 = N / n1;
        if (n2 > 4096 || n2 < 2) continue;
        if (!jit_cols_feasible(plan, n1, n2 * inner)) continue;
        DimPass b;
        std::string w;
        Plan probe = plan;
        probe.passes.clear();
        const bool fb = (make_cols_pass(probe, dim_index, n2, n1 * inner, b, w) && std::string(b.kernel_name) != "generic") ||
                        jit_cols_feasible(plan, n2, n1 * inner);
        if (!fb) continue;
        const double score = std::fabs(std::log((double)n1 / (double)n2));
        if (score < best_score) {
            best_score = score;
            best1 = n1;
            best2 = n2;
        }
    }
    if (!best1) {
        why_not = "no factorisation N1 * N2 with fused column tiles for both factors";
        return false;
    }
    DimPass pa, pb;
    pa.dim_index = dim_index;
    pa.N = best1;
    pa.inner = inner;
    pa.outer = outer;
    pa.fs_n1 = best1;
    pa.fs_n2 = best2;
    pa.first = false;
    if (!select_jit_fs1(plan, pa, why_not)) return false;
    pa.src_buf = 1;  // out
    pa.dst_buf = 2;  // scratch
    if (!make_cols_pass(plan, dim_index, best2, best1 * inner, pb, why_not, true)) return false;
    if (std::string(pb.kernel_name) == "generic") {
        why_not = "second factor only has a literal-stage kernel";
        return false;
    }
    pb.src_buf = 2;
    pb.dst_buf = 1;
    if (pb.prepare && pb.prepare() != MIFFT_OK) return false;
    const bool inv = plan.inverse != 0;
    hipError_t e = upload_twiddle_table(plan.out_dtype, pa.N, inv, &pa.d_twiddle);
    if (e == hipSuccess) e = upload_twiddle_table(plan.out_dtype, N, inv, &pa.d_aux);  // W_N^m for the k1 * n2 twiddle
    if (e == hipSuccess) e = upload_twiddle_table(plan.out_dtype, pb.N, inv, &pb.d_twiddle);
    if (e == hipSuccess) e = alloc_scratch(plan);
    if (e != hipSuccess) {
        free_pass_tables(pa);
        free_pass_tables(pb);
        why_not = std::string("device allocation: ") + hipGetErrorString(e);
        plan.alloc_failed = true;
        return false;
    }
    plan.passes.push_back(pa);
    plan.passes.push_back(pb);
    return true;
}

// ---------------------------------------------------------------------------------------------
// A STRIDED dimension too long for a column tile (N > 4096: 8K-video columns, ...): transpose the
// [N][inner] matrices into the plan scratch, run the contiguous-row kernel of length N there, transpose
// back.  Three passes, each fully coalesced -- the reference's own transpose / FFT / transpose scheme
// (fft/fft/_ndim_fft_gpu.mojo:634-642), used here only where an in-place column tile cannot be built.
// ---------------------------------------------------------------------------------------------
bool build_transposed_dim(Plan& plan, int dim_index, const std::vector<uint32_t>& radices,
                          const std::vector<uint32_t>& processed, std::string& why_not) {
    const int64_t N = plan.dims[dim_index];
    int64_t inner = 1, outer = 1;
    for (int k = dim_index + 1; k < plan.ndim; ++k) inner *= plan.dims[k];
    for (int k = 0; k < dim_index; ++k) outer *= plan.dims[k];
    if (inner == 1) {
        why_not = "contiguous dimension";
        return false;
    }
    DimPass rows;
    rows.dim_index = dim_index;
    rows.N = N;
    rows.inner = 1;
    rows.outer = outer * inner;  // rows of the transposed tensor per batch entry
    rows.radices = radices;
    rows.processed = processed;
    rows.first = false;
    std::string whyj;
    if (!select_fast(plan, rows) && !select_jit(plan, rows, whyj) && !select_generic(plan, rows, why_not)) return false;
    rows.src_buf = 2;
    rows.dst_buf = 2;
    DimPass t_in, t_out;
    t_in.dim_index = t_out.dim_index = dim_index;
    t_in.N = t_out.N = N;
    t_in.outer = t_out.outer = outer;
    t_in.kernel_name = t_out.kernel_name = "transpose";
    t_in.launch = t_out.launch = launch_transpose_twiddle;
    t_in.fs_n1 = N;       // out [N][inner] -> scratch [inner][N]
    t_in.fs_n2 = inner;
    t_in.src_buf = 1;
    t_in.dst_buf = 2;
    t_out.fs_n1 = inner;  // scratch [inner][N] -> out [N][inner]
    t_out.fs_n2 = N;
    t_out.src_buf = 2;
    t_out.dst_buf = 1;
    if (rows.prepare && rows.prepare() != MIFFT_OK) return false;
    // every allocation first; the three passes join the plan only when all of them succeeded (a plan that kept them
    // with a NULL scratch would write through a null pointer at the first exec)
    hipError_t e = upload_twiddle_table(plan.out_dtype, N, plan.inverse != 0, &rows.d_twiddle);
    if (e == hipSuccess) e = alloc_scratch(plan);
    if (e != hipSuccess) {
        free_pass_tables(rows);
        why_not = std::string("device allocation: ") + hipGetErrorString(e);
        plan.alloc_failed = true;
        return false;
    }
    plan.passes.push_back(t_in);
    plan.passes.push_back(rows);
    plan.passes.push_back(t_out);
    return true;
}

}  // namespace mifft
