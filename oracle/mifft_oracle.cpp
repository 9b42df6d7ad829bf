// mifft_oracle.cpp -- CPU restatement of the reference's radix-N FFT CPU path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the product (libmifft)
// never links, imports or calls it.
//
// Parity status: PINNED for convention and radix ordering by the reference's
// own golden vectors (tests/golden/*.json, extracted from fft/_test_values.mojo
// and fft/tests.mojo) and cross-checked against fp64 pocketfft; the reference
// itself (Mojo) cannot be built in this image, so bit-level fp32 rounding at
// the BASELINE sizes (N = 1024, 93, 640x480, 128^3) is "parity unpinned" by any
// reference-run output -- see DESIGN.md.
//
// Every function cites the reference file:line it follows (paths relative to
// the reference repository root).  Third-party arithmetic the reference uses
// that is not under /root/reference: Mojo stdlib `ComplexSIMD.fma`, `cos`,
// `sin` (mojo 0.26.3.0.dev2026032521, fft/pixi.lock:168-176) -- restated here
// as four real FMAs and libm cos/sin.
//
// Build: see oracle/Makefile (g++ -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp: std::fma becomes the hardware
// instruction instead of a libm call; contraction stays off, so FMAs appear exactly where the reference writes .fma()).

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

// Same numeric values as include/mifft.h (kept literal so the oracle has no
// build dependency on the product tree).
enum { ORA_F32 = 0, ORA_F64 = 1, ORA_U8 = 2, ORA_I32 = 3, ORA_I8 = 4, ORA_I16 = 5, ORA_U16 = 6, ORA_F16 = 7, ORA_BF16 = 8 };

// 16-bit floating inputs as storage types: the reference's `x.load(...).cast[out_dtype]()` (fft/fft/_fft.mojo:243-257)
// widens them exactly; g++ 11 has no _Float16 in C++, so the IEEE binary16 -> binary32 conversion is spelled out.
struct F16In {
    uint16_t b;
    operator float() const {
        const uint32_t sign = (uint32_t)(b & 0x8000u) << 16, e = (b >> 10) & 0x1Fu, m = b & 0x3FFu;
        uint32_t u;
        if (e == 0) {
            if (m == 0) {
                u = sign;
            } else {  // subnormal: normalise
                int sh = 0;
                uint32_t mm = m;
                while (!(mm & 0x400u)) {
                    mm <<= 1;
                    ++sh;
                }
                u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FFu) << 13);
            }
        } else if (e == 31) {
            u = sign | 0x7F800000u | (m << 13);
        } else {
            u = sign | ((e + 127 - 15) << 23) | (m << 13);
        }
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
};
struct BF16In {
    uint16_t b;
    operator float() const {
        const uint32_t u = (uint32_t)b << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
};
enum {
    ORA_OK = 0,
    ORA_ERR_BAD_RANK = -1,
    ORA_ERR_BAD_DIM = -2,
    ORA_ERR_BAD_COMPONENTS = -3,
    ORA_ERR_BAD_DTYPE = -4,
    ORA_ERR_BAD_BASES = -5,
    ORA_ERR_BASE_ONE = -6,
    ORA_ERR_NO_BASES = -7,
    ORA_ERR_BAD_BATCH = -8,
    ORA_ERR_NULL = -12,
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---------------------------------------------------------------------------
// planning utilities -- fft/fft/_utils.mojo:125-221, fft/fft/fft.mojo:49-104
// ---------------------------------------------------------------------------

// _div_by, fft/fft/_utils.mojo:125-129
static uint64_t div_by(uint64_t x, uint64_t base) {
    if (base == x) return 1;
    if (base > x || x % base != 0) return 0;
    return div_by(x / base, base) + 1;
}

// _times_divisible_by, fft/fft/_utils.mojo:132-152
static uint64_t times_divisible_by(uint64_t length, uint64_t base) {
    if (base != 0 && (base & (base - 1)) == 0) {  // power of two
        uint64_t tz = length == 0 ? 64 : (uint64_t)__builtin_ctzll(length);
        uint64_t lg = (uint64_t)__builtin_ctzll(base);  // log2(Float64(base)) cast to uint
        return lg == 0 ? 0 : tz / lg;
    }
    return div_by(length, base);
}

// _build_ordered_bases, fft/fft/_utils.mojo:162-183
static std::vector<uint64_t> build_ordered_bases(uint64_t length, std::vector<uint64_t> bases) {
    std::sort(bases.begin(), bases.end());
    uint64_t prod = 1;
    for (auto b : bases) prod *= b;
    if (prod == length) {
        std::reverse(bases.begin(), bases.end());
        return bases;
    }
    std::vector<uint64_t> out;
    uint64_t processed = 1;
    for (int i = (int)bases.size() - 1; i >= 0; --i) {
        uint64_t base = bases[i];
        uint64_t amnt = times_divisible_by(length, base);
        for (uint64_t k = 0; k < amnt; ++k) {
            out.push_back(base);
            processed *= base;
        }
        if (processed == length) break;
    }
    return out;
}

// _get_ordered_bases_processed_list, fft/fft/_utils.mojo:186-221
static int ordered_bases_processed(uint64_t length, const std::vector<uint64_t>& bases,
                                   std::vector<uint64_t>& ordered, std::vector<uint64_t>& processed) {
    if (bases.empty()) return fail(ORA_ERR_NO_BASES, "The amount of bases is not enough");
    for (auto b : bases)
        if (b <= 1) return fail(ORA_ERR_BASE_ONE, "Cannot do an fft with base 1.");
    ordered = build_ordered_bases(length, bases);
    processed.clear();
    uint64_t p = 1;
    for (auto b : ordered) {
        processed.push_back(p);
        p *= b;
    }
    if (ordered.empty() || p != length) {
        std::string s = "powers of the bases must multiply together to equal the sequence length. "
                        "The builtin algorithm was only able to produce: [";
        for (size_t i = 0; i < ordered.size(); ++i) s += (i ? ", " : "") + std::to_string(ordered[i]);
        s += "] for the length: " + std::to_string(length);
        return fail(ORA_ERR_BAD_BASES, s);
    }
    return ORA_OK;
}

// _estimate_best_bases, fft/fft/fft.mojo:49-104
static std::vector<uint64_t> estimate_best_bases(uint64_t length, bool gpu) {
    const uint64_t max_radix_number = 32, common_thread_block_size = 1024;
    if (gpu && length / max_radix_number <= common_thread_block_size) {
        uint64_t min_radix_for_block = (length + common_thread_block_size - 1) / common_thread_block_size;
        std::vector<uint64_t> potential;
        uint64_t processed = 1;
        for (uint64_t r = std::max<uint64_t>(min_radix_for_block, 2); r <= max_radix_number; ++r) {
            uint64_t amnt = times_divisible_by(length / processed, r);
            for (uint64_t k = 0; k < amnt; ++k) {
                potential.push_back(r);
                processed *= r;
            }
            if (processed == length) {
                std::reverse(potential.begin(), potential.end());
                return potential;
            }
        }
    }
    static const uint64_t primes[25] = {97, 89, 83, 79, 73, 71, 67, 61, 59, 53, 47, 43, 41,
                                        37, 31, 29, 23, 19, 17, 13, 11, 7,  5,  3,  2};
    std::vector<uint64_t> bases;
    uint64_t processed = 1;
    for (int i = 0; i < 25; ++i) {
        uint64_t amnt = times_divisible_by(length / processed, primes[i]);
        for (uint64_t k = 0; k < amnt; ++k) {
            bases.push_back(primes[i]);
            processed *= primes[i];
        }
        if (processed == length) {
            std::reverse(bases.begin(), bases.end());
            return bases;
        }
    }
    return bases;  // incomplete: the reference returns it as is (fft.mojo:88-104) and fails later
}

// ---------------------------------------------------------------------------
// arithmetic -- Mojo stdlib ComplexSIMD.fma restated; fft/fft/_utils.mojo:63-104
// ---------------------------------------------------------------------------

template <typename T>
struct Cx {
    T re, im;
};

// twf.fma(x, acc) = twf * x + acc, four real FMAs (call sites fft/fft/_fft.mojo:290,
// fft/fft/_utils.mojo:346).
template <typename T>
static inline Cx<T> cfma(Cx<T> w, Cx<T> x, Cx<T> c) {
    Cx<T> r;
    r.re = std::fma(w.re, x.re, -std::fma(w.im, x.im, -c.re));
    r.im = std::fma(w.re, x.im, std::fma(w.im, x.re, c.im));
    return r;
}

static inline float tcos(float x) { return cosf(x); }
static inline float tsin(float x) { return sinf(x); }
static inline double tcos(double x) { return cos(x); }
static inline double tsin(double x) { return sin(x); }

// _get_twiddle_factor, fft/fft/_utils.mojo:63-104.  `snap` = the
// __is_run_in_comptime_interpreter branch (:73-82), taken for compile-time tables.
template <typename T>
static Cx<T> twiddle_factor(uint64_t n, uint64_t N, bool inverse, bool snap) {
    const T c = (T)(-2.0 * M_PI) / (T)N;  // `-2π/N` evaluated in the working dtype (:68)
    volatile T theta = c * (T)n;          // :69
    Cx<T> num;
    bool done = false;
    if (snap) {
        T factor = (T)2 * (T)n / (T)N;
        if (factor < (T)1e-9) {
            num = {1, 0};
            done = true;
        } else if (factor == (T)0.5) {
            num = {0, -1};
            done = true;
        } else if (factor == (T)1) {
            num = {-1, 0};
            done = true;
        } else if (factor == (T)1.5) {
            num = {0, 1};
            done = true;
        }
    }
    if (!done) num = {tcos((T)theta), tsin((T)theta)};
    if (inverse) num.im = -num.im;  // .conj() (:101-104)
    return num;
}

// classification of a compile-time twiddle for _unit_phasor_fma (fft/fft/_utils.mojo:320-372)
enum : uint8_t { K_FMA = 0, K_ONE, K_MINUS_I, K_MINUS_ONE, K_PLUS_I, K_Q1, K_Q2, K_Q3, K_Q4 };

template <typename T>
static uint8_t classify(Cx<T> w) {
    if (w.re == (T)1) return K_ONE;
    if (w.im == (T)-1) return K_MINUS_I;
    if (w.re == (T)-1) return K_MINUS_ONE;
    if (w.im == (T)1) return K_PLUS_I;
    if (std::fabs(w.re) == std::fabs(w.im)) {
        if (w.re > 0 && w.im > 0) return K_Q1;
        if (w.re < 0 && w.im > 0) return K_Q2;
        if (w.re < 0 && w.im < 0) return K_Q3;
        return K_Q4;
    }
    return K_FMA;
}

// _unit_phasor_fma[twf](x_j, acc), fft/fft/_utils.mojo:320-346
template <typename T>
static inline Cx<T> unit_phasor_fma(uint8_t kind, Cx<T> w, Cx<T> x, Cx<T> acc) {
    switch (kind) {
        case K_ONE: return {acc.re + x.re, acc.im + x.im};
        case K_MINUS_I: return {acc.re + x.im, acc.im - x.re};
        case K_MINUS_ONE: return {acc.re - x.re, acc.im - x.im};
        case K_PLUS_I: return {acc.re - x.im, acc.im + x.re};
        case K_Q1: {
            T f = std::fabs(w.re);
            return {acc.re + f * (x.re - x.im), acc.im + f * (x.re + x.im)};
        }
        case K_Q2: {
            T f = std::fabs(w.re);
            return {acc.re + f * (-x.re - x.im), acc.im + f * (x.re - x.im)};
        }
        case K_Q3: {
            T f = std::fabs(w.re);
            return {acc.re + f * (-x.re + x.im), acc.im + f * (-x.re - x.im)};
        }
        case K_Q4: {
            T f = std::fabs(w.re);
            return {acc.re + f * (x.re + x.im), acc.im + f * (-x.re + x.im)};
        }
        default: return cfma(w, x, acc);
    }
}

// _unit_phasor_fma[twf, accum_is_real](x_j.re, acc), fft/fft/_utils.mojo:349-372
template <typename T>
static inline Cx<T> unit_phasor_fma_real(Cx<T> w, bool accum_is_real, T x, Cx<T> acc) {
    if (w.re == (T)1) return {acc.re + x, acc.im};
    if (w.im == (T)-1 && accum_is_real) return {acc.re, -x};
    if (w.im == (T)-1) return {acc.re, acc.im - x};
    if (w.re == (T)-1) return {acc.re - x, acc.im};
    if (w.im == (T)1 && accum_is_real) return {acc.re, x};
    if (w.im == (T)1) return {acc.re, acc.im + x};
    if (accum_is_real) return {std::fma(w.re, x, acc.re), w.im * x};
    return {std::fma(w.re, x, acc.re), std::fma(w.im, x, acc.im)};
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------

static const int MAX_STACK_SEQ_LEN = 128;  // fft/fft/_ndim_fft_cpu.mojo:34

template <typename T>
struct DimPlan {
    int64_t N = 0;
    std::vector<uint64_t> radices, processed;
    bool small = false;                 // N <= MAX_STACK_SEQ_LEN -> compile-time variant
    std::vector<Cx<T>> tw;              // length-N table W_N^n (snapped iff small)
    // index folding the reference does at compile time (small) or per element (large):
    std::vector<std::vector<int32_t>> src_n;   // [stage][i]            fft/fft/_fft.mojo:233-235
    std::vector<std::vector<int32_t>> tw_idx;  // [stage][(j-1)*N + i]  fft/fft/_fft.mojo:264-267
    std::vector<std::vector<uint8_t>> kind;    // small only
    // large N only: the stage's twiddles laid out [(j-1)][t], t = s*P + p = i % (P*R) -- the same values tw[tw_idx] reads,
    // contiguous along p so that the p loop of run_stage_rows vectorises
    std::vector<std::vector<Cx<T>>> tw_seq;
};

struct OraclePlan {
    int in_dtype, out_dtype, ndim, in_components, inverse;
    int64_t dims[6];
    int64_t batch, prod;
    std::vector<std::vector<uint64_t>> bases;  // user (or default) bases per dim
    void* dimplans[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // DimPlan<float|double>*
    std::vector<char> calc_buf;                 // _CPUPlan.calc_buf, _ndim_fft_cpu.mojo:44-45
    int64_t max_batch_prod = 0;                 // _find_max_batch_prod, :126-134
    int total_stages = 0;                       // :141-143
};

template <typename T>
static void build_dimplan(DimPlan<T>& dp, int64_t N, const std::vector<uint64_t>& ordered,
                          const std::vector<uint64_t>& processed, bool inverse) {
    dp.N = N;
    dp.radices = ordered;
    dp.processed = processed;
    dp.small = N <= MAX_STACK_SEQ_LEN;
    dp.tw.resize(N);
    // _get_twiddle_factors (runtime list, N > 128, _ndim_fft_cpu.mojo:58-60) or
    // _get_twiddle_factors_inline (compile-time, _fft.mojo:370-373)
    for (int64_t n = 0; n < N; ++n) dp.tw[n] = twiddle_factor<T>((uint64_t)n, (uint64_t)N, inverse, dp.small);
    size_t S = ordered.size();
    dp.src_n.resize(S);
    dp.tw_idx.resize(S);
    dp.kind.resize(S);
    dp.tw_seq.resize(S);
    for (size_t b = 0; b < S; ++b) {
        const int64_t R = (int64_t)ordered[b], P = (int64_t)processed[b];
        const int64_t next_offset = P * R, ratio = N / next_offset;
        dp.src_n[b].resize(N);
        dp.tw_idx[b].resize((R - 1) * N);
        if (dp.small) dp.kind[b].resize((R - 1) * N);
        for (int64_t i = 0; i < N; ++i) {
            dp.src_n[b][i] = (int32_t)((i / next_offset) * P + (i % next_offset) % P);
            for (int64_t j = 1; j < R; ++j) {
                int64_t base_idx = j * (i % next_offset);
                int64_t twf_index = (base_idx % next_offset) * ratio;
                dp.tw_idx[b][(j - 1) * N + i] = (int32_t)twf_index;
                if (dp.small) dp.kind[b][(j - 1) * N + i] = classify(dp.tw[twf_index]);
            }
        }
        if (!dp.small) {
            dp.tw_seq[b].resize((R - 1) * next_offset);
            for (int64_t j = 1; j < R; ++j)
                for (int64_t t = 0; t < next_offset; ++t) dp.tw_seq[b][(j - 1) * next_offset + t] = dp.tw[dp.tw_idx[b][(j - 1) * N + t]];
        }
    }
}

// ---------------------------------------------------------------------------
// one Stockham stage over one contiguous row
//   large N : _radix_n_fft_kernel_stockham          fft/fft/_fft.mojo:189-296
//   N <= 128: _radix_n_fft_kernel_stockham_comptime fft/fft/_fft.mojo:299-391
// ---------------------------------------------------------------------------

template <typename T, typename TIn>
static inline T cast_in(TIn v) {
    if constexpr (std::is_same<TIn, F16In>::value || std::is_same<TIn, BF16In>::value) return (T)(float)v;
    else return (T)v;
}

template <typename T, typename TIn>
static inline Cx<T> load_x(const TIn* x, int64_t idx, int comps) {
    if (comps == 1) return {cast_in<T>(x[idx]), (T)0};                   // _fft.mojo:254-255
    return {cast_in<T>(x[2 * idx]), cast_in<T>(x[2 * idx + 1])};          // _fft.mojo:257
}

// MODE: 0 large N (runtime twiddle table, plain complex FMA), 1 N <= 128 (strength-reduced phasors), 2 N <= 128 real
// input.  COMPS and MODE are template parameters only so that the compiler can drop the per-element branches; the
// arithmetic and its order are exactly those of the generic statement (the golden-vector tests pin them bit for bit).
template <typename T, typename TIn, int COMPS, int MODE>
static void run_stage_impl(const DimPlan<T>& dp, size_t b, Cx<T>* dst, const TIn* src, bool inverse) {
    const int64_t N = dp.N, R = (int64_t)dp.radices[b], P = (int64_t)dp.processed[b];
    const int64_t step = N / R;
    const bool last_inverse = inverse && P * R == N;
    const T inv_n = (T)(1.0 / (double)N);  // (1.0 / Float64(length)).cast[out_dtype]()
    const int32_t* sn = dp.src_n[b].data();
    const int32_t* ti = dp.tw_idx[b].data();
    const uint8_t* kd = MODE != 0 ? dp.kind[b].data() : nullptr;
    const Cx<T>* tw = dp.tw.data();
    for (int64_t i = 0; i < N; ++i) {
        const int64_t n = sn[i];
        Cx<T> acc = load_x<T, TIn>(src, n, COMPS);
        for (int64_t j = 1; j < R; ++j) {
            Cx<T> xj = load_x<T, TIn>(src, n + j * step, COMPS);
            Cx<T> w = tw[ti[(j - 1) * N + i]];
            if (MODE == 0) {
                acc = cfma(w, xj, acc);  // _fft.mojo:290
            } else if (MODE == 2) {
                acc = unit_phasor_fma_real(w, j == 1, xj.re, acc);  // _fft.mojo:382-383
            } else {
                acc = unit_phasor_fma(kd[(j - 1) * N + i], w, xj, acc);  // _fft.mojo:385
            }
        }
        if (last_inverse) {  // _fft.mojo:292-294 / :387-389
            acc.re *= inv_n;
            acc.im *= inv_n;
        }
        dst[i] = acc;
    }
}

// The large-N stage (MODE 0) on complex input of the working dtype, walked as q / s / p with p innermost: output
// i = q*P*R + s*P + p reads src[q*P + p + j*N/R] and the twiddle of t = s*P + p -- everything contiguous along p, which is
// how the reference's unrolled scalar loop (fft/fft/_ndim_fft_cpu.mojo:212-241, `vectorize[1, unroll_factor=width]` over
// local_i) ends up being scheduled.  Per output element the operations and their order are those of run_stage_impl
// (acc = x_0; acc = w_j.fma(x_j, acc) for j = 1..R-1; optional 1/N), so results are bit-identical; only the loop nest
// differs.
template <typename T>
static void run_stage_rows(const DimPlan<T>& dp, size_t b, Cx<T>* __restrict dst, const Cx<T>* __restrict src, bool inverse) {
    const int64_t N = dp.N, R = (int64_t)dp.radices[b], P = (int64_t)dp.processed[b];
    const int64_t step = N / R, PR = P * R;
    const bool last_inverse = inverse && PR == N;
    const T inv_n = (T)(1.0 / (double)N);
    const Cx<T>* __restrict tws = dp.tw_seq[b].data();
    for (int64_t q = 0; q < N / PR; ++q) {
        for (int64_t s = 0; s < R; ++s) {
            Cx<T>* __restrict d = dst + q * PR + s * P;
            const Cx<T>* __restrict x0 = src + q * P;
            for (int64_t p = 0; p < P; ++p) d[p] = x0[p];
            for (int64_t j = 1; j < R; ++j) {
                const Cx<T>* __restrict xj = x0 + j * step;
                const Cx<T>* __restrict w = tws + (j - 1) * PR + s * P;
                for (int64_t p = 0; p < P; ++p) d[p] = cfma(w[p], xj[p], d[p]);  // _fft.mojo:290
            }
            if (last_inverse)
                for (int64_t p = 0; p < P; ++p) {
                    d[p].re *= inv_n;
                    d[p].im *= inv_n;
                }
        }
    }
}

template <typename T, typename TIn>
static void run_stage(const DimPlan<T>& dp, size_t b, Cx<T>* dst, const TIn* src, int comps,
                      bool do_rfft, bool inverse) {
    const int mode = !dp.small ? 0 : (do_rfft ? 2 : 1);
    if constexpr (std::is_same<T, TIn>::value) {
        if (mode == 0 && comps == 2 && dp.processed[b] >= 8) {  // (shorter runs of p: the per-element loop below)
            run_stage_rows<T>(dp, b, dst, (const Cx<T>*)src, inverse);
            return;
        }
    }
    if (comps == 1) {
        if (mode == 0) run_stage_impl<T, TIn, 1, 0>(dp, b, dst, src, inverse);
        else if (mode == 1) run_stage_impl<T, TIn, 1, 1>(dp, b, dst, src, inverse);
        else run_stage_impl<T, TIn, 1, 2>(dp, b, dst, src, inverse);
    } else {
        if (mode == 0) run_stage_impl<T, TIn, 2, 0>(dp, b, dst, src, inverse);
        else if (mode == 1) run_stage_impl<T, TIn, 2, 1>(dp, b, dst, src, inverse);
        else run_stage_impl<T, TIn, 2, 2>(dp, b, dst, src, inverse);
    }
}

// _num_stages_end_of, fft/fft/_utils.mojo:379-397 (CPU: no scratch shortcut)
template <typename T>
static int num_stages_end_of(const OraclePlan& pl, int dim_idx) {
    int n = 0;
    for (int i = dim_idx; i < pl.ndim; ++i) n += (int)((DimPlan<T>*)pl.dimplans[i])->radices.size();
    return n;
}

// _run_1d_fft, fft/fft/_ndim_fft_cpu.mojo:145-241
template <typename T, typename TIn>
static void run_1d_fft(const OraclePlan& pl, int dim_idx, Cx<T>* lhs, Cx<T>* rhs, const TIn* x_in) {
    const DimPlan<T>& dp = *(const DimPlan<T>*)pl.dimplans[dim_idx];
    const int start_dim_idx = pl.ndim - 1;
    const int fft_stages = num_stages_end_of<T>(pl, dim_idx + 1);
    const int prev_stages = fft_stages + (start_dim_idx - dim_idx);
    for (size_t b = 0; b < dp.radices.size(); ++b) {
        const bool first = b == 0 && dim_idx == start_dim_idx;
        const bool do_rfft = pl.in_components == 1 && dim_idx == start_dim_idx && b == 0;
        const int s = prev_stages + (int)b;
        const bool write_lhs = (pl.total_stages - (s + 1)) % 2 == 0;
        Cx<T>* dst = write_lhs ? lhs : rhs;
        if (first) {
            run_stage<T, TIn>(dp, b, dst, x_in, pl.in_components, do_rfft, pl.inverse);
        } else {
            const Cx<T>* src = write_lhs ? rhs : lhs;
            run_stage<T, T>(dp, b, dst, (const T*)src, 2, false, pl.inverse);
        }
    }
}

// _transpose, fft/fft/_ndim_fft_cpu.mojo:63-93 with _calc_batches_M_N, _utils.mojo:400-419
template <typename T>
static void transpose_blocks(Cx<T>* dst, const Cx<T>* src, int64_t blocks, int64_t M, int64_t N,
                             int num_workers) {
    const int64_t TILE = 64 / (int64_t)sizeof(T) > 0 ? 64 / (int64_t)sizeof(T) : 1;  // simd_width_of
    int nw = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(num_workers, blocks), TILE));
#pragma omp parallel for num_threads(nw) schedule(static) if (nw > 1)
    for (int64_t b = 0; b < blocks; ++b) {
        const Cx<T>* s = src + b * M * N;
        Cx<T>* d = dst + b * M * N;
        for (int64_t i = 0; i < M; i += TILE)
            for (int64_t j = 0; j < N; j += TILE)
                for (int64_t ii = i; ii < std::min(i + TILE, M); ++ii)
                    for (int64_t jj = j; jj < std::min(j + TILE, N); ++jj) d[jj * M + ii] = s[ii * N + jj];
    }
}

// _run_batch, fft/fft/_ndim_fft_cpu.mojo:253-321
template <typename T, typename TIn>
static void run_batch(const OraclePlan& pl, int64_t block_num, Cx<T>* output, Cx<T>* calc, const TIn* x,
                      int per_batch_workers) {
    const int nd = pl.ndim, start = nd - 1;
    Cx<T>* base_out = output + block_num * pl.prod;
    Cx<T>* base_calc = calc + block_num * pl.prod;
    const TIn* base_x = x + block_num * pl.prod * pl.in_components;
    if (nd == 1) {
        run_1d_fft<T, TIn>(pl, start, base_out, base_calc, base_x);
        return;
    }
    for (int idx = nd - 1; idx >= 0; --idx) {
        const int64_t dim = pl.dims[idx];
        const int64_t batch_prod = pl.prod / dim;
        if (idx != start) {
            const int fft_stages = num_stages_end_of<T>(pl, idx + 1);
            const int s = fft_stages + (start - (idx + 1));
            const bool write_lhs = (pl.total_stages - (s + 1)) % 2 == 0;
            // _transpose[from_=idx+1, into_=idx]: (batch, M, N) = (prod dims[:idx], dims[idx], prod dims[idx+1:])
            int64_t bv = 1, nv = 1;
            for (int i = 0; i < idx; ++i) bv *= pl.dims[i];
            for (int i = idx + 1; i < nd; ++i) nv *= pl.dims[i];
            if (write_lhs)
                transpose_blocks<T>(base_out, base_calc, bv, dim, nv, per_batch_workers);
            else
                transpose_blocks<T>(base_calc, base_out, bv, dim, nv, per_batch_workers);
        }
#pragma omp parallel for num_threads(per_batch_workers) schedule(static) if (per_batch_workers > 1)
        for (int64_t flat = 0; flat < batch_prod; ++flat) {
            run_1d_fft<T, TIn>(pl, idx, base_out + flat * dim, base_calc + flat * dim,
                               base_x + flat * dim * pl.in_components);
        }
    }
    const int fft_stages = num_stages_end_of<T>(pl, 0);
    for (int idx = 0; idx < nd - 1; ++idx) {
        const int s = fft_stages + start + idx;
        const bool write_lhs = (pl.total_stages - (s + 1)) % 2 == 0;
        // _transpose[from_=idx, into_=idx+1]: (batch, M, N) = (prod dims[:idx], prod dims[idx+1:], dims[idx])
        int64_t bv = 1, nv = 1;
        for (int i = 0; i < idx; ++i) bv *= pl.dims[i];
        for (int i = idx + 1; i < nd; ++i) nv *= pl.dims[i];
        if (write_lhs)
            transpose_blocks<T>(base_out, base_calc, bv, nv, pl.dims[idx], per_batch_workers);
        else
            transpose_blocks<T>(base_calc, base_out, bv, nv, pl.dims[idx], per_batch_workers);
    }
}

// _run_cpu_nd_fft, fft/fft/_ndim_fft_cpu.mojo:96-323 (thread split :136-140, :323)
template <typename T, typename TIn>
static void run_nd(OraclePlan& pl, Cx<T>* output, const TIn* x, int64_t first, int64_t count, int cpu_workers) {
    int threads = cpu_workers;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_num_procs();  // parallelism_level()
    omp_set_max_active_levels(2);
#else
    threads = 1;
#endif
    int64_t per_batch_workers = pl.ndim > 1 ? std::min<int64_t>(threads, pl.max_batch_prod) : 1;
    int64_t parallel_batches =
        std::min<int64_t>(std::max<int64_t>(threads - (per_batch_workers - 1), 1), count);
    Cx<T>* calc = (Cx<T>*)pl.calc_buf.data();
    int pb = (int)std::max<int64_t>(parallel_batches, 1);
#pragma omp parallel for num_threads(pb) schedule(static) if (pb > 1)
    for (int64_t blk = first; blk < first + count; ++blk) {
        run_batch<T, TIn>(pl, blk, output, calc, x, (int)per_batch_workers);
    }
}

template <typename T>
static int dispatch_in(OraclePlan& pl, void* out, const void* x, int64_t first, int64_t count, int workers) {
    switch (pl.in_dtype) {
        case ORA_F32: run_nd<T, float>(pl, (Cx<T>*)out, (const float*)x, first, count, workers); break;
        case ORA_F64: run_nd<T, double>(pl, (Cx<T>*)out, (const double*)x, first, count, workers); break;
        case ORA_U8: run_nd<T, uint8_t>(pl, (Cx<T>*)out, (const uint8_t*)x, first, count, workers); break;
        case ORA_I32: run_nd<T, int32_t>(pl, (Cx<T>*)out, (const int32_t*)x, first, count, workers); break;
        case ORA_I8: run_nd<T, int8_t>(pl, (Cx<T>*)out, (const int8_t*)x, first, count, workers); break;
        case ORA_I16: run_nd<T, int16_t>(pl, (Cx<T>*)out, (const int16_t*)x, first, count, workers); break;
        case ORA_U16: run_nd<T, uint16_t>(pl, (Cx<T>*)out, (const uint16_t*)x, first, count, workers); break;
        case ORA_F16: run_nd<T, F16In>(pl, (Cx<T>*)out, (const F16In*)x, first, count, workers); break;
        case ORA_BF16: run_nd<T, BF16In>(pl, (Cx<T>*)out, (const BF16In*)x, first, count, workers); break;
        default: return fail(ORA_ERR_BAD_DTYPE, "unsupported in_dtype");
    }
    return ORA_OK;
}

// ---------------------------------------------------------------------------
// C ABI (mirrors include/mifft.h so tests read alike)
// ---------------------------------------------------------------------------
extern "C" {

const char* mifft_oracle_last_error(void) { return g_err.c_str(); }

int mifft_oracle_ordered_bases(uint32_t length, const uint32_t* bases, int nbases, uint32_t* out, int cap) {
    std::vector<uint64_t> b(bases, bases + (nbases > 0 ? nbases : 0)), ord, proc;
    int rc = ordered_bases_processed(length, b, ord, proc);
    if (rc) return rc;
    for (size_t i = 0; i < ord.size() && (int)i < cap; ++i) out[i] = (uint32_t)ord[i];
    return (int)ord.size();
}

int mifft_oracle_estimate_bases(uint32_t length, int target_gpu, uint32_t* out, int cap) {
    auto b = estimate_best_bases(length, target_gpu != 0);
    for (size_t i = 0; i < b.size() && (int)i < cap; ++i) out[i] = (uint32_t)b[i];
    return (int)b.size();
}

// plan_fft (CPU overload), fft/fft/fft.mojo:123-157 + _check_layout_conditions_nd :20-46
int mifft_oracle_plan_create(void** out_plan, int in_dtype, int out_dtype, int ndim, const int64_t* dims,
                             int64_t batch, int in_components, int inverse, const uint32_t* bases_flat,
                             const int32_t* bases_len, int default_target_gpu) {
    if (!out_plan || !dims) return fail(ORA_ERR_NULL, "null argument");
    *out_plan = nullptr;
    if (ndim < 1 || ndim > 6) return fail(ORA_ERR_BAD_RANK, "The rank should be bigger than 2 (ndim in 1..6)");
    if (in_components < 1 || in_components > 2)
        return fail(ORA_ERR_BAD_COMPONENTS, "The last dimension of in_layout should be 1 or 2");
    if (out_dtype != ORA_F32 && out_dtype != ORA_F64) return fail(ORA_ERR_BAD_DTYPE, "out_dtype must be floating point");
    if (in_dtype < ORA_F32 || in_dtype > ORA_BF16) return fail(ORA_ERR_BAD_DTYPE, "unsupported in_dtype");
    if (batch < 0) return fail(ORA_ERR_BAD_BATCH, "batch < 0");
    for (int i = 0; i < ndim; ++i)
        if (dims[i] < 2) return fail(ORA_ERR_BAD_DIM, "no inner dimension should be of size 1");

    OraclePlan* pl = new OraclePlan();
    pl->in_dtype = in_dtype;
    pl->out_dtype = out_dtype;
    pl->ndim = ndim;
    pl->in_components = in_components;
    pl->inverse = inverse ? 1 : 0;
    pl->batch = batch;
    pl->prod = 1;
    for (int i = 0; i < ndim; ++i) {
        pl->dims[i] = dims[i];
        pl->prod *= dims[i];
    }
    const uint32_t* bp = bases_flat;
    pl->total_stages = 2 * (ndim - 1);
    for (int i = 0; i < ndim; ++i) {
        std::vector<uint64_t> user;
        if (bases_flat && bases_len) {
            for (int k = 0; k < bases_len[i]; ++k) user.push_back(*bp++);
        } else {
            user = estimate_best_bases((uint64_t)dims[i], default_target_gpu != 0);
        }
        std::vector<uint64_t> ord, proc;
        int rc = ordered_bases_processed((uint64_t)dims[i], user, ord, proc);
        if (rc) {
            for (int k = 0; k < i; ++k) {
                if (out_dtype == ORA_F32) delete (DimPlan<float>*)pl->dimplans[k];
                else delete (DimPlan<double>*)pl->dimplans[k];
            }
            delete pl;
            return rc;
        }
        pl->bases.push_back(user);
        uint64_t mn = *std::min_element(user.begin(), user.end());
        pl->max_batch_prod = std::max<int64_t>(pl->max_batch_prod, (int64_t)((uint64_t)dims[i] / mn));
        pl->total_stages += (int)ord.size();
        if (out_dtype == ORA_F32) {
            auto* dp = new DimPlan<float>();
            build_dimplan<float>(*dp, dims[i], ord, proc, inverse != 0);
            pl->dimplans[i] = dp;
        } else {
            auto* dp = new DimPlan<double>();
            build_dimplan<double>(*dp, dims[i], ord, proc, inverse != 0);
            pl->dimplans[i] = dp;
        }
    }
    size_t esz = out_dtype == ORA_F32 ? 8 : 16;
    pl->calc_buf.resize((size_t)batch * (size_t)pl->prod * esz);
    *out_plan = pl;
    return ORA_OK;
}

// fft (CPU overload), fft/fft/fft.mojo:213-259
int mifft_oracle_exec_batch(void* plan, const void* x, void* out, int64_t first, int64_t count, int cpu_workers) {
    if (!plan || !x || !out) return fail(ORA_ERR_NULL, "null argument");
    OraclePlan& pl = *(OraclePlan*)plan;
    if (first < 0 || count < 0 || first + count > pl.batch) return fail(ORA_ERR_BAD_BATCH, "batch range");
    if (count == 0) return ORA_OK;
    if (pl.out_dtype == ORA_F32) return dispatch_in<float>(pl, out, x, first, count, cpu_workers);
    return dispatch_in<double>(pl, out, x, first, count, cpu_workers);
}

int mifft_oracle_exec(void* plan, const void* x, void* out, int cpu_workers) {
    if (!plan) return fail(ORA_ERR_NULL, "null plan");
    return mifft_oracle_exec_batch(plan, x, out, 0, ((OraclePlan*)plan)->batch, cpu_workers);
}

int mifft_oracle_plan_stages(void* plan, int dim, uint32_t* out, int cap) {
    if (!plan) return fail(ORA_ERR_NULL, "null plan");
    OraclePlan& pl = *(OraclePlan*)plan;
    if (dim < 0 || dim >= pl.ndim) return fail(ORA_ERR_BAD_RANK, "dim out of range");
    const std::vector<uint64_t>& r = pl.out_dtype == ORA_F32 ? ((DimPlan<float>*)pl.dimplans[dim])->radices
                                                              : ((DimPlan<double>*)pl.dimplans[dim])->radices;
    for (size_t i = 0; i < r.size() && (int)i < cap; ++i) out[i] = (uint32_t)r[i];
    return (int)r.size();
}

void mifft_oracle_plan_destroy(void* plan) {
    if (!plan) return;
    OraclePlan* pl = (OraclePlan*)plan;
    for (int k = 0; k < pl->ndim; ++k) {
        if (pl->out_dtype == ORA_F32) delete (DimPlan<float>*)pl->dimplans[k];
        else delete (DimPlan<double>*)pl->dimplans[k];
    }
    delete pl;
}

int mifft_oracle_num_procs(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

}  // extern "C"
