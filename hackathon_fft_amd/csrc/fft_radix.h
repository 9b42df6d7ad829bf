// fft_radix.h -- compile-time radix-R butterflies held entirely in registers (gfx950).
//
// One thread computes a whole forward R-point DFT, natural order in and out.  This is the
// "one thread computes a whole radix-R butterfly" formulation (the reference has it only as the
// dead Cooley-Tukey kernel, fft/fft/_fft.mojo:22-186; its live Stockham kernel
// fft/fft/_fft.mojo:189-296 spends one thread per OUTPUT and R-1 sequential complex FMAs).
// Mathematically each butterfly here is the same R-point DFT the reference's radix-R stage
// applies; a product of consecutive reference stages (e.g. 2*2*2*2) is evaluated as one
// composite butterfly (16).  Inverse transforms reuse these forward butterflies through
// conj(F(conj x)), which is bit-identical to running with conjugated twiddles.
#pragma once

// hipRTC (runtime specialisation, kernels_jit.cpp) brings the HIP device declarations itself; asking it for the header
// file makes the build depend on an include path that is not always there (under rocprofv3 it is not: the include
// failed and every runtime-specialised plan silently fell back to the literal-stage kernels)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

// Results must not depend on which tile slot / unrolled code instance a transform lands in (a slab of
// the batch must equal the same rows of the whole batch bit for bit, also across GPUs), so the compiler
// may not choose where to contract a*b+c: contraction is off and every FMA below is explicit.
#ifndef MIFFT_ALLOW_CONTRACT
#pragma clang fp contract(off)
#endif

namespace mifft {

#define MIFFT_DEV __host__ __device__ __forceinline__

MIFFT_DEV float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MIFFT_DEV double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};

// bfloat16 as an INPUT element type (MIFFT_BF16): the upper half of a binary32, widened exactly
struct bf16_t {
    unsigned short bits;
    MIFFT_DEV operator float() const { return __builtin_bit_cast(float, (unsigned)bits << 16); }
};

template <typename T> MIFFT_DEV cpx<T> operator+(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> MIFFT_DEV cpx<T> operator-(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> MIFFT_DEV cpx<T> cmul(cpx<T> a, cpx<T> b) {
#ifdef MIFFT_EXPLICIT_FMA
    return {fma_t(a.x, b.x, -(a.y * b.y)), fma_t(a.x, b.y, a.y * b.x)};
#else
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
#endif
}
template <typename T> MIFFT_DEV cpx<T> mul_neg_i(cpx<T> a) { return {a.y, -a.x}; }  // a * (-i)
template <typename T> MIFFT_DEV cpx<T> mul_pos_i(cpx<T> a) { return {-a.y, a.x}; }  // a * (+i)

// ---- constexpr trigonometry (exact octant reduction + Taylor on [0, pi/4]) ----
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_small(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int k = 1; k <= 11; ++k) {
        term *= -x2 / ((2.0 * k) * (2.0 * k + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double cx_cos_small(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k <= 11; ++k) {
        term *= -x2 / ((2.0 * k - 1.0) * (2.0 * k));
        sum += term;
    }
    return sum;
}
struct cx_pair {
    double c, s;
};
// (cos, sin) of 2*pi*num/den, num may be any integer
constexpr cx_pair cx_cossin(long long num, long long den) {
    num %= den;
    if (num < 0) num += den;
    // f = num/den in [0,1)
    bool neg_sin = false, neg_cos = false, swap = false;
    long long n = num, d = den;
    if (2 * n > d) {  // f > 1/2 : angle -> 2pi - angle
        n = d - n;
        neg_sin = true;
    }
    if (4 * n > d) {  // f > 1/4 : angle -> pi - angle
        n = d - 2 * n;  // (1/2 - f) = (d - 2n)/(2d)
        d = 2 * d;
        neg_cos = true;
    }
    if (8 * n > d) {  // f > 1/8 : angle -> pi/2 - angle
        n = d - 4 * n;  // (1/4 - f) = (d - 4n)/(4d)
        d = 4 * d;
        swap = true;
    }
    double x = 2.0 * kPi * (double)n / (double)d;
    double c = cx_cos_small(x), s = cx_sin_small(x);
    if (swap) {
        double t = c;
        c = s;
        s = t;
    }
    if (neg_cos) c = -c;
    if (neg_sin) s = -s;
    return {c, s};
}

// v * W_DEN^NUM  (forward twiddle exp(-2*pi*i*NUM/DEN)), special angles strength-reduced
template <int NUM_, int DEN, typename T>
MIFFT_DEV cpx<T> mul_w(cpx<T> v) {
    constexpr int NUM = ((NUM_ % DEN) + DEN) % DEN;
    if constexpr (NUM == 0) {
        return v;
    } else if constexpr (4 * NUM == DEN) {
        return mul_neg_i(v);
    } else if constexpr (2 * NUM == DEN) {
        return {-v.x, -v.y};
    } else if constexpr (4 * NUM == 3 * DEN) {
        return mul_pos_i(v);
    } else if constexpr (8 * NUM == DEN) {  // (1 - i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210485;
        return {h * (v.x + v.y), h * (v.y - v.x)};
    } else if constexpr (8 * NUM == 3 * DEN) {  // (-1 - i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210485;
        return {h * (v.y - v.x), -h * (v.x + v.y)};
    } else if constexpr (8 * NUM == 5 * DEN) {  // (-1 + i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210485;
        return {-h * (v.x + v.y), h * (v.x - v.y)};
    } else if constexpr (8 * NUM == 7 * DEN) {  // (1 + i)/sqrt2
        constexpr T h = (T)0.70710678118654752440084436210485;
        return {h * (v.x - v.y), h * (v.x + v.y)};
    } else {
        constexpr cx_pair cs = cx_cossin(NUM, DEN);
        constexpr T c = (T)cs.c, s = (T)(-cs.s);  // exp(-i a) = cos a - i sin a
#ifdef MIFFT_EXPLICIT_FMA
        return {fma_t(v.x, c, -(v.y * s)), fma_t(v.x, s, v.y * c)};
#else
        return {v.x * c - v.y * s, v.x * s + v.y * c};
#endif
    }
}

constexpr bool is_prime_ce(int n) {
    if (n < 2) return false;
    for (int d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}
// split factor A for composite R = A * B (Cooley-Tukey inside registers)
constexpr int split_factor(int r) {
    if (r % 4 == 0 && r > 4) return (r == 8) ? 2 : 4;
    for (int d = 2; d * d <= r; ++d)
        if (r % d == 0) return d;
    return r;
}

template <int R, typename T, int STRIDE = 1>
struct Dft;

// ---- R = 2, 4 ----
template <typename T, int S>
struct Dft<2, T, S> {
    static MIFFT_DEV void run(cpx<T>* v) {
        cpx<T> a = v[0], b = v[S];
        v[0] = a + b;
        v[S] = a - b;
    }
};
template <typename T, int S>
struct Dft<4, T, S> {
    static MIFFT_DEV void run(cpx<T>* v) {
        cpx<T> a0 = v[0] + v[2 * S], a1 = v[0] - v[2 * S];
        cpx<T> a2 = v[S] + v[3 * S], a3 = mul_neg_i(v[S] - v[3 * S]);
        v[0] = a0 + a2;
        v[S] = a1 + a3;
        v[2 * S] = a0 - a2;
        v[3 * S] = a1 - a3;
    }
};

// ---- odd prime R: conjugate-pair (symmetric) form, ~4x fewer multiplies than the O(R^2)
//      complex-FMA chain of the reference stage (fft/fft/_fft.mojo:261-290) ----
template <int R, typename T, int S>
struct DftOddPrime {
    static constexpr int H = (R - 1) / 2;
    template <int s, int j>
    static MIFFT_DEV void acc(const cpx<T>* a, const cpx<T>* b, cpx<T>& A, cpx<T>& B) {
        if constexpr (j <= H) {
            constexpr cx_pair cs = cx_cossin((long long)j * s, R);
            constexpr T c = (T)cs.c, sn = (T)cs.s;
#ifndef MIFFT_NO_PRIME_FMA
            A.x = fma_t(c, a[j - 1].x, A.x);
            A.y = fma_t(c, a[j - 1].y, A.y);
            B.x = fma_t(sn, b[j - 1].x, B.x);
            B.y = fma_t(sn, b[j - 1].y, B.y);
#else
            A.x += c * a[j - 1].x;
            A.y += c * a[j - 1].y;
            B.x += sn * b[j - 1].x;
            B.y += sn * b[j - 1].y;
#endif
            acc<s, j + 1>(a, b, A, B);
        }
    }
    template <int s>
    static MIFFT_DEV void outputs(cpx<T>* v, cpx<T> x0, const cpx<T>* a, const cpx<T>* b) {
        if constexpr (s <= H) {
            cpx<T> A = x0, B = {(T)0, (T)0};
            acc<s, 1>(a, b, A, B);
            // X_s = A - i B ; X_{R-s} = A + i B
            v[s * S] = {A.x + B.y, A.y - B.x};
            v[(R - s) * S] = {A.x - B.y, A.y + B.x};
            outputs<s + 1>(v, x0, a, b);
        }
    }
    // EMIT-AS-YOU-GO form: the sums a_j / differences b_j overwrite the inputs in place (v[j] <- a_j, v[R-j] <- b_j) and
    // every conjugate output pair is handed to `emit(s, X_s)` the moment it exists, so the R outputs never live in
    // registers beside the R - 1 sums (radix 31: ~70 live VGPRs instead of ~130).  Same operations in the same order as
    // run(): results are bit-identical.
    template <int s, class Emit>
    static MIFFT_DEV void emit_pairs(const cpx<T>* v, cpx<T> x0, Emit& emit) {
        if constexpr (s <= H) {
            cpx<T> A = x0, B = {(T)0, (T)0};
            acc_inplace<s, 1>(v, A, B);
            emit(s, cpx<T>{A.x + B.y, A.y - B.x});
            emit(R - s, cpx<T>{A.x - B.y, A.y + B.x});
            emit_pairs<s + 1>(v, x0, emit);
        }
    }
    template <int s, int j>
    static MIFFT_DEV void acc_inplace(const cpx<T>* v, cpx<T>& A, cpx<T>& B) {  // a_j at v[j * S], b_j at v[(R - j) * S]
        if constexpr (j <= H) {
            constexpr cx_pair cs = cx_cossin((long long)j * s, R);
            constexpr T c = (T)cs.c, sn = (T)cs.s;
#ifndef MIFFT_NO_PRIME_FMA
            A.x = fma_t(c, v[j * S].x, A.x);
            A.y = fma_t(c, v[j * S].y, A.y);
            B.x = fma_t(sn, v[(R - j) * S].x, B.x);
            B.y = fma_t(sn, v[(R - j) * S].y, B.y);
#else
            A.x += c * v[j * S].x;
            A.y += c * v[j * S].y;
            B.x += sn * v[(R - j) * S].x;
            B.y += sn * v[(R - j) * S].y;
#endif
            acc_inplace<s, j + 1>(v, A, B);
        }
    }
    template <class Emit>
    static MIFFT_DEV void run_emit(cpx<T>* v, Emit& emit) {
        const cpx<T> x0 = v[0];
        cpx<T> sum = x0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const cpx<T> u = v[j * S], w = v[(R - j) * S];
            v[j * S] = u + w;
            v[(R - j) * S] = u - w;
        }
#pragma unroll
        for (int j = 1; j <= H; ++j) sum = sum + v[j * S];
        emit(0, sum);
        emit_pairs<1>(v, x0, emit);
    }
    static MIFFT_DEV void run(cpx<T>* v) {
#ifdef MIFFT_ABLATE_PRIME_DFT  // timing experiment only: how fast is the kernel around a free butterfly?
        if (R > 16) return;
#endif
        cpx<T> a[H], b[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            a[j - 1] = v[j * S] + v[(R - j) * S];
            b[j - 1] = v[j * S] - v[(R - j) * S];
        }
        cpx<T> x0 = v[0], sum = v[0];
#pragma unroll
        for (int j = 0; j < H; ++j) sum = sum + a[j];
        v[0] = sum;
        outputs<1>(v, x0, a, b);
    }
};

// ---- odd prime R, OUTPUT-SPLIT form: lane group g of G computes only the conjugate pairs
//      s in [g*PP + 1, (g+1)*PP] (PP = ceil(H / G)) -- group 0 also X_0 -- and hands every result to
//      `emit(s, value)` at once, so nothing but the a/b sums stays live.  Used where the butterfly count
//      per row is tiny (93 = 31 * 3 has three) to put G lanes on one butterfly. ----
template <int R, typename T, int G, int g>
struct PrimeGroup {
    static constexpr int H = (R - 1) / 2, PP = (H + G - 1) / G, S0 = g * PP + 1,
                         S1 = (S0 + PP - 1 < H) ? (S0 + PP - 1) : H;
    template <int s, class Emit>
    static MIFFT_DEV void pairs(cpx<T> x0, const cpx<T>* a, const cpx<T>* b, Emit& emit) {
        if constexpr (s <= S1) {
            cpx<T> A = x0, B = {(T)0, (T)0};
            DftOddPrime<R, T, 1>::template acc<s, 1>(a, b, A, B);
            emit(s, cpx<T>{A.x + B.y, A.y - B.x});
            emit(R - s, cpx<T>{A.x - B.y, A.y + B.x});
            pairs<s + 1>(x0, a, b, emit);
        }
    }
    template <class Emit>
    static MIFFT_DEV void run(const cpx<T>* x, Emit& emit) {
        cpx<T> a[H], b[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            a[j - 1] = x[j] + x[R - j];
            b[j - 1] = x[j] - x[R - j];
        }
        if constexpr (g == 0) {
            cpx<T> sum = x[0];
#pragma unroll
            for (int j = 0; j < H; ++j) sum = sum + a[j];
            emit(0, sum);
        }
        pairs<S0>(x[0], a, b, emit);
    }
};

// ---- composite R = A * B: B-point DFTs over the residue classes mod A, twiddle, A-point DFTs ----
//   x[A m + r]  --DFT_B over m-->  Y_r[k2]  --* W_R^{r k2}-->  --DFT_A over r-->  X[k2 + B k1]
template <int R, typename T, int S>
struct DftComposite {
    static constexpr int A = split_factor(R), B = R / A;
    template <int r, int k2>
    static MIFFT_DEV void twiddle(cpx<T>* y) {
        if constexpr (r < A) {
            if constexpr (k2 < B) {
                y[r * B + k2] = mul_w<r * k2, R>(y[r * B + k2]);
                twiddle<r, k2 + 1>(y);
            } else {
                twiddle<r + 1, 1>(y);
            }
        }
    }
    static MIFFT_DEV void run(cpx<T>* v) {
        cpx<T> y[R];  // y[r*B + m]
#pragma unroll
        for (int r = 0; r < A; ++r)
#pragma unroll
            for (int m = 0; m < B; ++m) y[r * B + m] = v[(A * m + r) * S];
#pragma unroll
        for (int r = 0; r < A; ++r) Dft<B, T, 1>::run(y + r * B);  // Y_r[k2] at y[r*B + k2]
        twiddle<1, 1>(y);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) Dft<A, T, B>::run(y + k2);  // over r (stride B): X[k2 + B k1] at y[k1*B + k2]
#pragma unroll
        for (int k = 0; k < R; ++k) v[k * S] = y[k];  // y[k1*B + k2] is X[k2 + B*k1] = X[k]
    }
};

template <int R, typename T, int S>
struct Dft {
    static MIFFT_DEV void run(cpx<T>* v) {
        if constexpr (is_prime_ce(R))
            DftOddPrime<R, T, S>::run(v);
        else
            DftComposite<R, T, S>::run(v);
    }
};

}  // namespace mifft
