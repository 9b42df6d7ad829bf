// tile_kernel.h -- register-butterfly Stockham tile kernel template for MI355X (gfx950).
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
// Twiddles: W_N^n from the plan's fp64-accurate table.  Per-thread twiddles are loop
// invariant over tiles, so a persistent workgroup keeps them in registers (TWMODE 1) or in a
// compact conflict-free LDS table [pass][j][p] (TWMODE 2); TWMODE 0 reads the global table.
// Inverse: conj(F(conj x)) * 1/N with the forward butterflies (bit-identical to conjugated
// twiddles), reference semantics fft/fft/_utils.mojo:101-104, fft/fft/_fft.mojo:292-294.
//
// Real input, N-D plans (the reference computes the whole spectrum in every pass): the LAST pass may transform only the
// columns that are not beyond their mirror image and store every result twice (TileCfg::HERM: contiguous runs of tiles, a
// carried column so that mirrored lines leave whole); the pass in front of it stores only the half it reads (TileCfg::HS),
// or -- three-pass plans -- the first pass does and the middle pass transforms a column prefix (TileParams::col_lim).
// TileCfg::R2C (packed real rows) is a measured tie kept for the lab build.  Selection and policy: mifft_api.cpp, herm_pays().
#pragma once

#include "fft_radix.h"

// cooperative prime pass (bigprime_pass): consecutive output pairs per thread (each input pair is read once per SB outputs)
#ifndef MIFFT_BIGP_SB
#define MIFFT_BIGP_SB 4
#endif

// odd prime radices from this size on use the emit-as-you-go butterfly in LDS-bound passes (pass_compute_scatter): radix 31
// drops from 165 to 145 VGPRs (1023 = 31 * 11 * 3: 0.069 -> 0.055 ms per 128 MB, rows93 real input stops spilling); radices
// 17 / 19 keep all their outputs in registers anyway and measured 0-7 % slower with it (867 = 17 * 17 * 3)
#ifndef MIFFT_EMIT_PRIME_MIN
#define MIFFT_EMIT_PRIME_MIN 23
#endif

namespace mifft {

template <typename T> struct native_vec2;
template <> struct native_vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct native_vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <bool NT, typename T>
MIFFT_DEV cpx<T> gload(const cpx<T>* p) {
    if constexpr (NT) {
        typename native_vec2<T>::type v = __builtin_nontemporal_load((const typename native_vec2<T>::type*)p);
        return {v.x, v.y};
    } else {
        return *p;
    }
}
template <bool NT, typename T>
MIFFT_DEV T gload_real(const T* p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}
template <bool NT, typename T>
MIFFT_DEV void gstore(cpx<T>* p, cpx<T> v) {
    if constexpr (NT) {
        typename native_vec2<T>::type w = {v.x, v.y};
        __builtin_nontemporal_store(w, (typename native_vec2<T>::type*)p);
    } else {
        *p = v;
    }
}

template <class A, class B> struct same_t { static constexpr bool value = false; };
template <class A> struct same_t<A, A> { static constexpr bool value = true; };

// one input element of a FOREIGN element type IT (uint8 / int8 / int16 / uint16 / int32 / half / bfloat16, or float under a
// double plan), real or interleaved complex, widened to cpx<T> -- the reference's `x.load(...).cast[out_dtype]()`,
// fft/fft/_fft.mojo:243-257
template <class C>
MIFFT_DEV cpx<typename C::T> load_foreign(const void* in, long long idx) {
    using T = typename C::T;
    using IT = typename C::IT;
    const IT* p = (const IT*)in;
    // through float for bf16_t (its only conversion); exact for every element type narrower than T
    auto widen = [](IT v) -> T {
        if constexpr (same_t<IT, bf16_t>::value) return (T)(float)v;
        else return (T)v;
    };
    if constexpr (C::IN_REAL) {
        return {widen(p[idx]), (T)0};
    } else {
        struct alignas(2 * sizeof(IT)) pair_t { IT re, im; };
        const pair_t v = ((const pair_t*)p)[idx];
        return {widen(v.re), widen(v.im)};
    }
}

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
    // TSTORE configurations (first pass of the four-step): two-level table of W_M^m, M = N * inner,
    // W_M^m = tlo[m mod 1024] * thi[m div 1024]
    const void* tlo;
    const void* thi;
    const void* tcol;  // [TILE][N]: W_M^(c * k1), the twiddle of column c0 + c relative to column c0
    // walk the tiles from the last to the first.  An in-place pass that follows a pass over the same tensor then
    // starts with the lines written LAST, i.e. the ones most likely to be still in the 256-MiB Infinity Cache (LRU:
    // walking forward again would evict exactly the lines it is about to need).
    int reverse;
    // FS1 configurations: `inner` is the LOAD row stride N2 * fs_inner; rows are stored at stride fs_inner
    long long fs_inner;
    long long fs_n2;
    // HERM configurations (last pass of a REAL-input N-D plan, in place): herm_d0 x herm_d1 x herm_d2 are the trailing
    // dimensions, i.e. the column space (leading ones 1 when absent).  One of them is the HALF AXIS (size herm_dj, stride
    // herm_js columns): the pass transforms the columns whose index along it is at most herm_dj / 2 and owns those below the
    // middle.  The column space is herm_rows rows of herm_L = herm_dj * herm_js columns; the first herm_H = (herm_dj / 2 + 1)
    // * herm_js columns of every row are covered by herm_tpr tiles, tiles_per_outer = rows * herm_tpr.
    int herm_d0, herm_d1, herm_d2;
    int herm_dj, herm_js, herm_L, herm_H, herm_tpr;
    // HS configurations (the pass BEFORE a HERM last pass; a plane: its column side): results with an output index above
    // store_lim are not stored -- the last pass reads only the half of that dimension up to its middle and writes the rest
    int store_lim;
    // R2C configurations: W_(2N)^k, k = 0 .. N (forward, whatever the plan's direction), and the row pitch of `out` (2 N)
    const void* r2c_tw;
    long long out_pitch;
    // column tiles of a pass that transforms only a PREFIX of its column space (the middle pass of a half-spectrum schedule):
    // columns from col_lim on are neither loaded nor stored (0 = all `inner` columns)
    long long col_lim;
};

MIFFT_DEV long long tile_id(const TileParams& p, long long t) { return p.reverse ? p.n_tiles - 1 - t : t; }

constexpr int ilog2_ce(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}
constexpr bool is_pow2_ce(int v) { return v > 0 && (v & (v - 1)) == 0; }

enum { TW_GLOBAL = 0, TW_REG = 1, TW_LDS = 2 };

// ---- Rader: a prime radix R too large for a register butterfly (R > 32) as a cyclic convolution of length M = R - 1 ----
// x[g^q] (*) W_R^(g^-q), computed with an M-point FFT, a pointwise product with the precomputed spectrum of the kernel and
// an inverse M-point FFT, all inside the LDS tile (rader_pass).  2 * 5 M log2 M flops per R-point DFT instead of the 4 (R/2)^2
// FMAs of the cooperative conjugate-pair pass (bigprime_pass): 3x fewer at R = 97, 20x at R = 1009.  Applicable when
// M = R - 1 splits into at most four register butterflies (factors <= 16, or one prime <= 31 each): rader_ok, evaluated by
// the host, which selects the pass per configuration (TileCfg::RADERM) and uploads its tables.
struct RaderSplitT {
    int np;
    int r[4];
    bool ok;
};
constexpr RaderSplitT rader_split(int M) {
    RaderSplitT s{0, {1, 1, 1, 1}, false};
    int pf[32] = {};
    int npf = 0, m = M;
    for (int d = 2; d <= m; ++d)
        while (m % d == 0) {
            if (npf >= 32) return s;
            pf[npf++] = d;
            m /= d;
        }
    bool used[32] = {};
    int left = npf;
    while (left > 0) {
        if (s.np >= 4) return s;
        int r = 1;
        for (int i = npf - 1; i >= 0; --i) {
            if (used[i]) continue;
            if (r == 1) {
                if (pf[i] > 31) return s;
                r = pf[i];
            } else if (r * pf[i] <= 16) {
                r *= pf[i];
            } else {
                continue;
            }
            used[i] = true;
            --left;
        }
        s.r[s.np++] = r;
    }
    s.ok = s.np >= 1;
    return s;
}
constexpr bool rader_ok(int R) { return R > 32 && rader_split(R - 1).ok; }
// R - 1 with a prime factor above 31 (83, 509, 2039 ...): the cyclic convolution of length M = R - 1 is embedded in one of
// length L >= 2 M - 1 (zero-padded sequence, wrapped kernel) with L a product of factors <= 16; it runs in a scratch block
// of L elements per DFT beside the data tile.  0 = none below 8192.
constexpr bool rader_small_radices(int L) {
    const RaderSplitT s = rader_split(L);
    if (!s.ok) return false;
    for (int i = 0; i < s.np; ++i)
        if (s.r[i] > 16) return false;
    return true;
}
constexpr int rader_pad_len(int R) {
    for (int L = 2 * (R - 1) - 1; L <= 8192; ++L)
        if (rader_small_radices(L)) return L;
    return 0;
}
// convolution length of a Rader pass: M in place, or the padded L
constexpr int rader_len(int R) { return rader_ok(R) ? R - 1 : rader_pad_len(R); }
// LDS of one Rader pass in units of one complex element of `esz` bytes: spectrum of the kernel [M], W_M [M], the two
// permutations (2 M uint16) and x_0 of every R-point DFT of the tile [inst]
// (+ the scratch blocks of a padded pass: inst * L)
constexpr int rader_lds_elems(int R, int inst, int esz) {
    const int L = rader_len(R);
    return 2 * L + (4 * (R - 1) + esz - 1) / esz + inst + (rader_ok(R) ? 0 : inst * L);
}

template <typename T_, int N_, int NP_, int R0_, int R1_, int R2_, int R3_, int TILE_, int THREADS_, bool COLS_,
          bool FIRST_DIRECT_, bool LAST_DIRECT_, int TWMODE_, int MINW_ = 1, bool PREFETCH_ = false, int ROWPAD_ = 0,
          bool IN_REAL_ = false, bool DMA_ = false, int NT_ = 0, bool TSTORE_ = false, typename IT_ = T_, bool WSUB_ = false,
          bool FS1_ = false, int RADERM_ = 0, bool HERM_ = false, bool HS_ = false, bool R2C_ = false>
struct TileCfg {
    using T = T_;
    static constexpr int N = N_, NP = NP_, TILE = TILE_, THREADS = THREADS_, TWMODE = TWMODE_, MINW = MINW_;
    static constexpr bool COLS = COLS_, FIRST_DIRECT = FIRST_DIRECT_, LAST_DIRECT = LAST_DIRECT_;
    static constexpr bool PREFETCH = PREFETCH_ && FIRST_DIRECT_;
    // pass 0 reads a REAL tensor (C_in = 1) and promotes it (fft/fft/_fft.mojo:254-255).  Compile-time:
    // a runtime switch inside the prefetching load loop cost 0.30 -> 0.49 ms at 100k x 1024.
    static constexpr bool IN_REAL = IN_REAL_;
    // element type of the tensor pass 0 reads (the plan's in_dtype): T itself, or uint8 / int32 / float widened to T
    // in the load (the reference's `x.load(...).cast[out_dtype]`, fft/fft/_fft.mojo:243-251).  Only the
    // runtime-specialised kernels (kernels_jit.cpp) instantiate foreign input types.
    using IT = IT_;
    // non-temporal HBM accesses (bit 0: loads, bit 1: stores).  A tile is read once and written once; the
    // streaming hint lifts a row-shaped copy from ~5.85 to ~6.4 TB/s on MI355X (tools/micro/copy_nt.hip) when
    // the tensors dwarf the 256-MB Infinity Cache, and HURTS cache-resident or in-place passes (measured:
    // N = 93 flat copy 0.173 -> 0.247 ms, 640-point column tiles 0.114 -> 0.139 ms), so it is a twin
    // configuration chosen at plan time by tensor size.
    static constexpr int NT = NT_;
    static constexpr int LD = N_ + ROWPAD_;  // ROWS: LDS pitch of one transform
    static constexpr int R(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : i == 2 ? R2_ : R3_; }
    static constexpr int P(int i) {
        int p = 1;
        for (int k = 0; k < i; ++k) p *= R(k);
        return p;
    }
    static constexpr int NB(int i) { return N / R(i); }
    static constexpr int ITEMS(int i) { return NB(i) * TILE; }
    // WSUB (column tiles): after pass 0 (radix R0) the transform falls apart into R0 independent sub-problems -- the
    // elements at positions = p (mod R0) -- and every later pass works inside them.  With WSUB a wave OWNS SPW of these
    // sub-problems (all TILE columns) for the passes 1..NP-1, so the exchanges between those passes touch only LDS
    // positions of its own sub-problems: one wave's LDS instructions execute in order and no workgroup barrier is
    // needed.  Two workgroup barriers per tile remain (after the pass-0 scatter, after the last gather) instead of
    // 2 (NP - 1); the waves drift apart and overlap each other's butterflies, LDS traffic and HBM stores.
    static constexpr bool WSUB = WSUB_;
    static constexpr int WAVES = THREADS_ / 64;
    static constexpr int SPW = WSUB_ ? R0_ / (WAVES > 0 ? WAVES : 1) : 1;  // sub-problems per wave
    static constexpr int WSLOTS = 64 / (TILE_ <= 64 ? TILE_ : 64);        // (sub-problem, butterfly) pairs per wave instruction
    static constexpr int WPER(int i) { return SPW * (NB(i) / R0_); }      // ... per wave and pass
    static_assert(!WSUB_ || (COLS_ && THREADS_ % 64 == 0 && 64 % TILE_ == 0 && R0_ % (THREADS_ / 64) == 0 && !TSTORE_ &&
                             TWMODE_ != TW_REG && NP_ >= 2 && FIRST_DIRECT_ && LAST_DIRECT_ && R0_ <= 32),
                  "WSUB: column tile, TILE divides a wave, R0 a multiple of the wave count");
    static constexpr int IPT(int i) {
        return (WSUB_ && i >= 1) ? (WPER(i) + WSLOTS - 1) / WSLOTS : (ITEMS(i) + THREADS - 1) / THREADS;
    }
    static constexpr int TW_OFF(int i) {  // register twiddles of passes 1..i-1 precede pass i
        int o = 0;
        for (int k = 1; k < i; ++k) o += IPT(k) * (R(k) - 1);
        return o;
    }
    static constexpr int TW_TOTAL = TW_OFF(NP_);
    static constexpr int TWL_OFF(int i) {  // compact LDS table [pass][j-1][p]
        int o = 0;
        for (int k = 1; k < i; ++k) o += P(k) * (R(k) - 1);
        return o;
    }
    static constexpr int TWL_TOTAL = TWMODE_ == TW_LDS ? TWL_OFF(NP_) : 0;
    // TSTORE (column tiles only): the finished tile is stored TRANSPOSED and multiplied by the four-step twiddle
    // W^{k1*n2}: columns c0..c0+TILE-1 of the [N][inner] matrix become TILE contiguous rows of the [inner][N]
    // matrix, so the store is one flat coalesced run.  The LDS column pitch is TILE + 1 so that the transposed
    // read (lanes along n) is conflict-free.
    // Re-derive every thread-index-dependent offset inside each tile iteration instead of letting the compiler
    // hoist them out of the persistent loop (see tile_kernel).  Column tiles and very long rows carry dozens of
    // such offsets and spill or lose occupancy when they stay live; the short-row kernels run at copy speed with
    // them hoisted and lose 3-8 % to the recomputation (A/B on MI355X, gpurun_out/ab_opaque.log).
#ifdef MIFFT_OPAQUE_ROWS
    static constexpr bool OPAQUE_TID = true;
#else
    static constexpr bool OPAQUE_TID = COLS_ || N_ >= 8192;
#endif
    // Tile -> workgroup mapping.  Workgroups are dealt round-robin to the 8 XCDs (workgroup id mod 8), each with its
    // own L2.  Column tiles whose runs are NARROWER than a 128-B line (long strided dimensions: 8 / 4 columns) give
    // every XCD one CONTIGUOUS eighth of the tiles, so that the two / four tiles sharing a line meet in one L2 instead
    // of each XCD fetching and writing the line partially: 8-column tiles gain 10 %, 4-column tiles 24-36 % (4K frame
    // 0.104 -> 0.078 ms).  Full-line tiles (16 columns) are mixed under the same mapping -- 640 / 128 / 256-point tiles
    // gain 3-7 %, the 2^20 four-step and 64^3 lose 6 % (same-box A/B) -- and row tiles are neutral, so
    // both keep the plain round-robin order.
#ifdef MIFFT_NO_XCD_CHUNK
    static constexpr bool XCD_CHUNK = false;
#else
    static constexpr bool XCD_CHUNK = COLS_ && TILE_ * 2 * (int)sizeof(T_) < 128;
#endif
    static constexpr bool TSTORE = TSTORE_;
    // FS1 (column tiles): first pass of the four-step of a STRIDED dimension of N = N1 * N2 points (N1 = this
    // configuration's N).  The dimension is viewed as [N1][N2] rows of `fs_inner` contiguous elements; the tile of
    // (n2, 16 columns) is transformed over n1 (row stride N2 * fs_inner) and its row k1 is stored, multiplied by
    // W_N^(k1 * n2), as row n2 * N1 + k1 of the OTHER buffer -- a row-granular transposition, so the stores keep their
    // TILE-element runs and need no LDS round trip.  The second pass (an ordinary column tile of N2 points at row
    // stride N1 * fs_inner) then leaves the spectrum in natural order.
    static constexpr bool FS1 = FS1_;
    // HERM (column tiles, in place): the LAST pass of a real-input N-D transform.  The spectrum of a real tensor is
    // Hermitian, Y[-k, -c] = conj(Y[k, c]) over the transformed dimensions, and so is every intermediate over the dimensions
    // already transformed: column -c of this pass holds the conjugate of column c.  Only the columns c <= -c (a flat prefix
    // of the column space) are read and transformed; every result is stored twice, at (k, c) and conjugated at (-k, -c).
    // Half the reads and butterflies of the pass; the reference computes (and this library's other kernels compute) all of it.
    static constexpr bool HERM = HERM_;
    static_assert(!HERM_ || (COLS_ && LAST_DIRECT_ && !TSTORE_ && !FS1_), "HERM: a direct in-place column tile");
    // HS ("half store"): the pass that transforms the dimension a following HERM pass halves -- the one before the last of a
    // real-input N-D plan -- stores only the results 0 .. store_lim (= N / 2) of every transform: the HERM pass reads nothing
    // else and writes every other point itself.  Half the writes of that pass.
    static constexpr bool HS = HS_;
    static_assert(!HS_ || (!TSTORE_ && !FS1_ && !HERM_), "HS: a plain row / column tile or the column side of a plane");
    // R2C: a row tile over a REAL tensor whose rows of 2 N points are read as N packed complex points z_n = x_2n + i x_2n+1
    // (the row as it lies in memory), transformed by the ordinary passes, and unpacked into X[0 .. N] of the 2 N-point real
    // transform by the store loop (the classic real-FFT split): half the butterflies and LDS traffic of a promoted (x, 0)
    // row.  Only the half spectrum is produced, so it serves where a Hermitian last pass follows (it is a half-store kernel).
    static constexpr bool R2C = R2C_;
    static_assert(!R2C_ || (!COLS_ && !LAST_DIRECT_ && !IN_REAL_ && !TSTORE_ && !FS1_ && !HERM_ && !HS_ && !DMA_),
                  "R2C: a row tile that leaves its last pass in LDS");
    // inverse plans run conj(F(conj x)): complex input is conjugated as it is loaded (a real input is its own conjugate; the
    // packed rows of R2C are unpacked first and conjugated as they are stored)
    static constexpr bool CONJ_IN = !IN_REAL_ && !R2C_;
    static_assert(!FS1_ || (COLS_ && FIRST_DIRECT_ && LAST_DIRECT_ && !TSTORE_ && !WSUB_), "FS1: a direct column tile");
    static constexpr int CPITCH = TSTORE_ ? TILE_ + 1 : TILE_;
    static_assert(!TSTORE_ || (COLS_ && !LAST_DIRECT_ && FIRST_DIRECT_), "TSTORE: column tile, last pass left in LDS");
    static constexpr int DATA_ELEMS = COLS_ ? N_ * CPITCH : LD * TILE_;
    // BIGP0: the first radix is a prime too large for a register butterfly (> 32).  Pass 0 then runs cooperatively in
    // LDS (bigprime_pass0): the row is staged by the flat copy, paired into sums / differences in place, and every
    // thread accumulates a few output pairs over the (R-1)/2 pairs with a cos/sin table of R entries kept behind the
    // twiddle table.  Instantiated by the runtime-specialised kernels only.
    static constexpr bool BIGP0 = R0_ > 32;
    // a second prime above 32 is pass 1 (its inputs are twiddled while they are paired)
    static constexpr bool BIGP1 = NP_ > 1 && R1_ > 32;
    static_assert(!(R2_ > 32 || R3_ > 32) && (!BIGP1 || BIGP0), "cooperative passes come first, at most two");
    static constexpr bool BIGP(int i) { return i == 0 ? BIGP0 : i == 1 ? BIGP1 : false; }
    // per cooperative pass: the cos/sin table of R entries (bigprime_pass), or the Rader tables (rader_pass)
    // (RADERM: bit i = cooperative pass i runs as a Rader convolution; chosen by the host -- it pays from R ~ 128 on and
    //  needs LDS for its tables -- and uploaded tables go with it)
    static constexpr bool RADER(int i) { return BIGP(i) && ((RADERM_ >> i) & 1) != 0; }
    static_assert(RADERM_ == 0 || ((!(RADERM_ & 1) || rader_len(R0_) > 0) && (!(RADERM_ & 2) || rader_len(R1_) > 0)),
                  "Rader pass: no convolution length that splits into register butterflies");
    static constexpr int cs_size(int i) {
        return !BIGP(i) ? 0 : RADER(i) ? rader_lds_elems(R(i), TILE_ * NB(i), 2 * (int)sizeof(T_)) : R(i);
    }
    // (data members: evaluated once at compile time -- a runtime call of rader_len factorises thousands of integers)
    static constexpr int CS_SIZE0 = cs_size(0), CS_SIZE1 = cs_size(1);
    static constexpr int CS_OFF(int i) { return i == 0 ? 0 : CS_SIZE0; }
    static constexpr int CS_ELEMS = CS_SIZE0 + CS_SIZE1;
    static_assert(!BIGP0 || (!FIRST_DIRECT_ && TWMODE_ == TW_LDS), "big-prime pass 0: tile staged in LDS first");
    // DMA: the flat HBM -> LDS copy of the NEXT tile runs asynchronously (global_load_lds) into a staging
    // buffer behind the twiddle table while this tile's passes execute
    static constexpr bool DMA = DMA_;
    static constexpr int STAGE_OFF = ((DATA_ELEMS + TWL_TOTAL) * 2 * (int)sizeof(T_) + 15) / 16 * 16 / (2 * (int)sizeof(T_));
    static constexpr int STAGE_ELEMS = DMA_ ? N_ * TILE_ : 0;
    // HERM: one column of N results carried from a tile to the next one of the workgroup's run (see the HERM stores)
    // (full-line tiles only: narrower ones need the XCD_CHUNK order above -- their neighbours have to run at the same time
    //  on one XCD, or every line is fetched once per tile that shares it: 2- and 4-column tiles 1.2-1.6x slower in runs)
    static constexpr bool HERM_RUNS = HERM_ && !XCD_CHUNK;
    static constexpr int HERM_OFF = DATA_ELEMS + TWL_TOTAL + CS_ELEMS;
    static constexpr int HERM_ELEMS = HERM_RUNS ? N_ : 0;
    static_assert(!(HERM_ && DMA_), "HERM: no staging buffer");
    static constexpr size_t LDS_BYTES =
        DMA_ ? (size_t)(STAGE_OFF + STAGE_ELEMS) * 2 * sizeof(T_)
             : (size_t)(DATA_ELEMS + TWL_TOTAL + CS_ELEMS + HERM_ELEMS) * 2 * sizeof(T_);
    static_assert(P(NP_) == N_, "radices must multiply to N");
    static_assert(LDS_BYTES <= 160 * 1024, "tile + twiddle table exceed the CU's 160 KiB of LDS");
};

// XOR swizzle of the in-row index for the exchange written by pass E (power-of-two rows
// only): the 16 lanes of a ds_write_b64 group own 16 butterflies whose outputs are P*R
// elements apart; fold the low butterfly bits into the bank-selecting low 4 index bits.
template <class C, int E>
MIFFT_DEV int swz(int n) {
    if constexpr (!C::COLS && is_pow2_ce(C::N) && E >= 0 && E < C::NP - 1) {
        // LG = log2(lanes per ds_write group): 16 for 8-byte elements (b64), 8 for 16-byte (b128)
        constexpr int LG = sizeof(typename C::T) == 4 ? 4 : 3;
        constexpr int a = ilog2_ce(C::P(E)), c = ilog2_ce(C::P(E) * C::R(E));
        constexpr int c4 = c < LG ? c : LG, hi = c > LG ? c : LG, nb = c4 - a;
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
        return n * C::CPITCH + c;
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

// Addressing split for HBM accesses: address = (uniform tile pointer + uniform element step) + 32-bit lane
// offset.  The uniform part lives in SGPRs, each butterfly carries ONE offset VGPR for all of its R accesses
// (global_load v, v_off, s[base] form) instead of R 64-bit address pairs.
template <class C>
MIFFT_DEV unsigned lane_off(const TileParams& p, int c, int n) {  // n: element index inside the transform
    if constexpr (C::COLS)
        return (unsigned)n * (unsigned)p.inner + (unsigned)c;
    else
        return (unsigned)c * (unsigned)C::N + (unsigned)n;
}
template <class C>
MIFFT_DEV long long elem_stride(const TileParams& p) {
    if constexpr (C::COLS)
        return p.inner;
    else
        return 1;
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

// work item k of thread tid in pass I -> (transform c of the tile, butterfly b); false: no such item
template <class C, int I>
MIFFT_DEV bool item_of(int tid, int k, int& c, int& b) {
    if constexpr (C::WSUB && I >= 1) {
        const int wave = tid >> 6, lane = tid & 63;
        const int m = k * C::WSLOTS + lane / C::TILE;  // pair (local sub-problem, butterfly inside it)
        c = lane % C::TILE;
        const int pl = m % C::SPW, qq = m / C::SPW;
        b = qq * C::R(0) + wave * C::SPW + pl;  // b mod R0 = the sub-problem
        return (C::WPER(I) % C::WSLOTS == 0) || m < C::WPER(I);
    } else {
        const int id = tid + k * C::THREADS;
        item_decode<C, I>(id, c, b);
        return (C::ITEMS(I) % C::THREADS == 0) || id < C::ITEMS(I);
    }
}

// order one wave's LDS accesses across a pass boundary for the COMPILER (the hardware executes one wave's LDS
// instructions in order; there is nothing to wait for)
// every vector-memory operation of this wave issued so far is complete (s_waitcnt vmcnt(0), visible to the compiler's
// own wait-count bookkeeping) and nothing scheduled after it moves above
MIFFT_DEV void vm_drain() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt unconstrained
    __builtin_amdgcn_sched_barrier(0);
}

MIFFT_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// FS1: tile t = (outer o, n2, column tile): load base / store base / twiddle row n2
template <class C>
MIFFT_DEV void tile_geom_fs1(const TileParams& p, long long t, long long& base, long long& obase, int& nv, int& n2) {
    const long long tiles_per_row = (p.fs_inner + C::TILE - 1) / C::TILE;
    const long long o = t / p.tiles_per_outer, r = t - o * p.tiles_per_outer;
    const long long row = r / tiles_per_row, c0 = (r - row * tiles_per_row) * C::TILE;
    const long long left = p.fs_inner - c0;
    nv = (int)(left < C::TILE ? left : C::TILE);
    n2 = (int)row;
    const long long image = o * (long long)C::N * p.inner;  // N1 * (N2 * fs_inner) elements per outer index
    base = image + row * p.fs_inner + c0;
    obase = image + row * (long long)C::N * p.fs_inner + c0;
}

template <class C>
MIFFT_DEV void tile_geom(const TileParams& p, long long t, long long& base, int& nv) {
    if constexpr (C::FS1) {
        long long obase;
        int n2;
        tile_geom_fs1<C>(p, t, base, obase, nv, n2);
    } else if constexpr (C::COLS) {
        const long long o = t / p.tiles_per_outer;
        const long long c0 = (t - o * p.tiles_per_outer) * C::TILE;
        const long long left = (p.col_lim ? p.col_lim : p.inner) - c0;
        nv = (int)(left < C::TILE ? left : C::TILE);
        base = o * (long long)C::N * p.inner + c0;
    } else {
        const long long r0 = t * C::TILE;
        const long long left = p.n_rows - r0;
        nv = (int)(left < C::TILE ? left : C::TILE);
        base = r0 * C::N;
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
MIFFT_DEV void fill_lds_tw(cpx<typename C::T>* ltw, const cpx<typename C::T>* tw, int tid, int inverse) {
    if constexpr (I < C::NP) {
        constexpr int R = C::R(I), P = C::P(I), RATIO = C::N / (P * R);
        for (int e = tid; e < P * (R - 1); e += C::THREADS) {
            const int j = e / P + 1, pp = e - (j - 1) * P;
            cpx<typename C::T> w = tw[j * pp * RATIO];
            if (inverse) w.y = -w.y;
            ltw[C::TWL_OFF(I) + e] = w;
        }
        fill_lds_tw<C, I + 1>(ltw, tw, tid, inverse);
    }
}

// Experiment switch: keep the butterflies of one thread apart in the instruction schedule (lower register pressure,
// less ILP).  Off in the product build.
#ifdef MIFFT_SCHED_FENCE
#define MIFFT_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MIFFT_FENCE() ((void)0)
#endif

// In-kernel phase stamps -- DIAGNOSTIC BUILDS ONLY (tools/tune, -DMIFFT_STAMPS; no product kernel executes one).
// Thread 0 of every workgroup accumulates s_memtime deltas per phase in 17 LDS words BEHIND the kernel's dynamic LDS
// (the tuner launches with 256 extra bytes) and copies them to the buffer passed in TileParams::tcol at the end
// ([workgroup][16] shader cycles); stamp values never reach an output element.
#ifdef MIFFT_STAMPS
#ifndef MIFFT_STAMP_TID
#define MIFFT_STAMP_TID 0  // the stamping thread (its wave is the one measured)
#endif
MIFFT_DEV void mifft_stamp(int i, unsigned lds_off, const void* dump = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stamp[];  // the kernel's dynamic LDS
    if (threadIdx.x == MIFFT_STAMP_TID) {
        unsigned long long* a = (unsigned long long*)(smem_stamp + ((lds_off + 15) / 16) * 16);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (i < 0) {
            for (int k = 0; k < 16; ++k) a[k] = 0;
        } else if (i < 16) {
            a[i] += t - a[16];
        } else {
            for (int k = 0; k < 16; ++k) ((unsigned long long*)dump)[(size_t)blockIdx.x * 16 + k] = a[k];
        }
        a[16] = __builtin_amdgcn_s_memtime();
    }
}
#define MIFFT_STAMP(C_, i) mifft_stamp(i, (unsigned)C_::LDS_BYTES)
#define MIFFT_STAMP_DUMP(C_, p_) mifft_stamp(99, (unsigned)C_::LDS_BYTES, (p_).tcol)
#else
#define MIFFT_STAMP(C_, i) ((void)0)
#define MIFFT_STAMP_DUMP(C_, p_) ((void)0)
#endif

// gather the pass-0 inputs of tile (base, nv) from HBM into registers; slice PART of NPARTS: the elements e = k R + j
// with e % NPARTS == PART (NPARTS = 1: all of them)
template <class C, int PART = 0, int NPARTS = 1>
MIFFT_DEV void load_pass0(const TileParams& p, cpx<typename C::T> (*v)[C::R(0)], long long base, int nv, int tid) {
    using T = typename C::T;
    using V = cpx<T>;
    constexpr int R = C::R(0), NB = C::NB(0), IPT = C::IPT(0);
    constexpr bool EXACT = C::ITEMS(0) % C::THREADS == 0;
    const V* gin = (const V*)p.in;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int id = tid + k * C::THREADS;
        if (EXACT || id < C::ITEMS(0)) {
            int c, b;
            item_decode<C, 0>(id, c, b);
            // No predicate on the loads: a slot beyond the ragged end of the last tile re-reads the last valid
            // transform (always in bounds; its results are never stored).  Branch-free loads stay back to back
            // with a single wait at first use -- per-load exec-masked blocks cost 40 % on the column tiles.
            const int cc = c < nv ? c : nv - 1;
            const unsigned off = lane_off<C>(p, cc, b);
            const long long step = (long long)NB * elem_stride<C>(p);  // uniform: element j sits j*step further
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if ((k * R + j) % NPARTS != PART) continue;  // (compile-time after unrolling)
                if constexpr (!same_t<typename C::IT, T>::value) {
                    v[k][j] = load_foreign<C>(p.in, base + j * step + off);
                } else if constexpr (C::IN_REAL) {
                    v[k][j].x = gload_real<(C::NT & 1) != 0>((const T*)p.in + base + j * step + off);
                    v[k][j].y = (T)0;
                } else {
                    v[k][j] = gload<(C::NT & 1) != 0>(gin + base + j * step + off);
                }
            }
        }
    }
}

// workgroup barrier.  Kernels with an LDS-DMA in flight (C::DMA) must not use __syncthreads(): its fence
// drains vmcnt and with it the asynchronous copy (cdna_hip_programming.md, "Pipelining across barriers").
template <class C>
MIFFT_DEV void wg_barrier() {
#ifdef MIFFT_ABLATE_BARRIERS  // timing experiment only (results are wrong): what do the workgroup barriers cost?
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return;
#endif
    if constexpr (C::DMA)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
        __syncthreads();
}

// LDS -> registers: the R inputs of every butterfly this thread owns in pass I, twiddled
template <class C, int I>
MIFFT_DEV void pass_gather_lds(const TileParams& p, const cpx<typename C::T>* src, const cpx<typename C::T>* ltw,
                               const cpx<typename C::T>* twr, cpx<typename C::T> (*v)[C::R(I)], int tid) {
    using T = typename C::T;
    using V = cpx<T>;
    constexpr int R = C::R(I), P = C::P(I), NB = C::NB(I), IPT = C::IPT(I), RATIO = C::N / (P * R);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int c, b;
        if (item_of<C, I>(tid, k, c, b)) {
#pragma unroll
            for (int j = 0; j < R; ++j) v[k][j] = src[lds_index<C, I - 1>(c, b + j * NB)];
            if constexpr (I > 0) {
                const int pp = b % P;
#pragma unroll
                for (int j = 1; j < R; ++j) {
                    V w;
                    if constexpr (C::TWMODE == TW_REG) {
                        w = twr[C::TW_OFF(I) + k * (R - 1) + (j - 1)];
                    } else if constexpr (C::TWMODE == TW_LDS) {
                        w = ltw[C::TWL_OFF(I) + (j - 1) * P + pp];
                    } else {
                        w = ((const V*)p.tw)[j * pp * RATIO];
                        if (p.inverse) w.y = -w.y;
                    }
#ifndef MIFFT_ABLATE_MATH
                    v[k][j] = cmul(v[k][j], w);
#else
                    v[k][j].x += w.x;
#endif
                }
            }
        }
        MIFFT_FENCE();
    }
}

// HERM: the flat column (kz, ky, kx) of the trailing dimensions (herm_d0 x herm_d1 x herm_d2) mirrors to (-kz, -ky, -kx)
MIFFT_DEV int herm_mirror(const TileParams& p, int cf) {
    const int d2 = p.herm_d2, d1 = p.herm_d1;
    const int r = cf / d2, kx = cf - r * d2, kz = r / d1, ky = r - kz * d1;
    return ((kz ? p.herm_d0 - kz : 0) * d1 + (ky ? d1 - ky : 0)) * d2 + (kx ? d2 - kx : 0);
}

// HERM: does this pass compute column cf itself (`own`), and has the column a mirror image other than itself (`twice`)?
// Below the middle of the half axis: yes; beyond it: no (the mirror image's tile stores it); on a self-mirrored index of the
// half axis the flat order of the pair decides.
MIFFT_DEV void herm_owner(const TileParams& p, int cf, int mf, bool& own, bool& twice) {
    const int in_row = cf % p.herm_L, kj = in_row / p.herm_js;
    own = (kj == 0 || 2 * kj == p.herm_dj) ? cf <= mf : 2 * kj < p.herm_dj;
    twice = own && cf != mf;
}
MIFFT_DEV bool herm_twice(const TileParams& p, int cf) {
    bool own, twice;
    herm_owner(p, cf, herm_mirror(p, cf), own, twice);
    return twice;
}
// HERM: tile t -> (outer index, row of the column space, tile inside the covered part of the row)
template <class C>
MIFFT_DEV void tile_geom_herm(const TileParams& p, long long t, long long& base, int& nv, int& col0, int& in_row) {
    const long long o = t / p.tiles_per_outer;
    const int in_image = (int)(t - o * p.tiles_per_outer);
    const int r = in_image / p.herm_tpr;
    in_row = in_image - r * p.herm_tpr;
    col0 = r * p.herm_L + in_row * C::TILE;
    const int left = p.herm_H - in_row * C::TILE;
    nv = left < C::TILE ? left : C::TILE;
    base = o * (long long)C::N * p.inner + col0;
}

// registers -> butterflies -> Stockham scatter (LDS, or HBM for the last pass)
template <class C, int I>
MIFFT_DEV void pass_compute_scatter(const TileParams& p, cpx<typename C::T>* lds, cpx<typename C::T> (*v)[C::R(I)],
                                    long long base, int nv, int tid, long long obase = 0, int fs_row = 0) {
    using T = typename C::T;
    using V = cpx<T>;
    constexpr int R = C::R(I), P = C::P(I), IPT = C::IPT(I);
    constexpr bool DST_GLOBAL = (I == C::NP - 1) && C::LAST_DIRECT;
    V* gout = (V*)p.out;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int c, b;
        if (item_of<C, I>(tid, k, c, b)) {
            const int q = b / P, pp = b - q * P;
            const int o0 = q * P * R + pp;
            // large odd primes scatter every conjugate pair to LDS the moment it is computed (DftOddPrime::run_emit): the
            // R outputs never exist in registers beside the R - 1 sums
            if constexpr (!DST_GLOBAL && is_prime_ce(R) && R >= MIFFT_EMIT_PRIME_MIN) {
                auto emit = [&](int s, V y) { lds[lds_index<C, I>(c, o0 + s * P)] = y; };
                DftOddPrime<R, T, 1>::run_emit(v[k], emit);
                MIFFT_FENCE();
                continue;
            }
#ifndef MIFFT_ABLATE_MATH  // timing experiment only: data movement without butterflies / twiddles
            Dft<R, T, 1>::run(v[k]);
#endif
            if constexpr (DST_GLOBAL && C::FS1) {
                if (c < nv) {
                    // row k1 = o0 + s * P of this tile becomes row n2 * N1 + k1 of the other buffer, times W^(k1 * n2)
                    const unsigned off = (unsigned)o0 * (unsigned)p.fs_inner + (unsigned)c;
                    const long long step = (long long)P * p.fs_inner;  // uniform
                    const V* wtab = (const V*)p.tlo;                    // W_(N1 N2)^m, m < N1 * N2
#pragma unroll
                    for (int s = 0; s < R; ++s) {
                        V w = wtab[(o0 + s * P) * fs_row];
                        if (p.inverse) w.y = -w.y;  // the plan's table is conjugated for inverse plans; forward W here
                        V y = cmul(v[k][s], w);
                        if (p.inverse) {
                            y.x *= (T)p.scale;
                            y.y *= -(T)p.scale;
                        }
                        gstore<(C::NT & 2) != 0>(gout + obase + s * step + off, y);
                    }
                }
            } else if constexpr (DST_GLOBAL && C::HERM) {
                // Every result goes to (k, cf) and, conjugated, to the mirrored point (-k, mf).  The mirror image of an
                // aligned run of TILE columns [c0, c0 + TILE) is the run [m - TILE + 1, m], which starts ONE element past a
                // line boundary: stored as it stands, every mirrored line would be written by two tiles in two pieces
                // (15 + 1 elements for 16-column tiles), each store instruction touching twice the lines -- measured 13-15 %
                // of the whole transform (DESIGN_EXPERIMENTS.md R3.6).  So a workgroup walks a CONTIGUOUS run of tiles in
                // DESCENDING column order and the lane of a tile's first column does not store its own mirror image: it
                // keeps the column in LDS (`carry`, N elements) and stores the column the previous tile kept, which is the
                // missing first element of this tile's mirrored lines.  Only the ends of a run store single elements.
                //   obase = the image's base (outer index); fs_row = the tile's first flat column (bits 0..29),
                //   bit 30: the previous tile of the run left its first column in `carry`, bit 31: leave this tile's there
                const int col0 = fs_row & 0x3fffffff;
                const bool carry_in = (fs_row >> 30) & 1, carry_out = ((unsigned)fs_row >> 31) != 0;
                const int cf = col0 + c, mf = herm_mirror(p, cf);
                bool own, has_mirror;
                herm_owner(p, cf, mf, own, has_mirror);
                const bool alive = c < nv && own;  // (the other columns belong to their mirror image's tile)
                const bool first = c == 0, twice = c < nv && has_mirror;
                const bool keep = first && twice && carry_out;              // first column: left in LDS for the next tile,
                const bool lone = first && twice && !carry_out && carry_in;  // or stored by itself at the end of a run
                const bool mirrored = first ? carry_in || (twice && !carry_out) : twice;           // the mirrored store ...
                const int mcol = first && carry_in ? herm_mirror(p, col0 + C::TILE) : mf;  // ... and its column
                V* carry = lds + C::HERM_OFF;
                const unsigned off = lane_off<C>(p, c, o0);
                const long long step = (long long)P * elem_stride<C>(p);  // uniform
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    V y = v[k][s];
                    if (p.inverse) {
                        y.x *= (T)p.scale;
                        y.y *= -(T)p.scale;
                    }
                    if (alive) gstore<(C::NT & 2) != 0>(gout + base + s * step + off, y);
                    const int kk = o0 + s * P, kr = kk ? C::N - kk : 0;
                    V* const mrow = gout + obase + (long long)kr * p.inner;
                    V z = y;
                    // (the thread that reads carry[kk] here is the one that wrote it in the previous tile: no barrier)
                    if constexpr (C::HERM_RUNS) {
                        if (first && carry_in) z = carry[kk];
                        if (keep) carry[kk] = y;
                    }
                    if (lone) {
                        V w = y;
                        w.y = -w.y;
                        gstore<(C::NT & 2) != 0>(mrow + mf, w);
                    }
                    z.y = -z.y;
                    if (mirrored) gstore<(C::NT & 2) != 0>(mrow + mcol, z);
                }
            } else if constexpr (DST_GLOBAL) {
                if (c < nv) {
                    const unsigned off = lane_off<C>(p, c, o0);
                    const long long step = (long long)P * elem_stride<C>(p);  // uniform
#pragma unroll
                    for (int s = 0; s < R; ++s) {
                        // (HS: store_lim is N / 2 and this is the last pass, o0 < P -- outputs s * P > N / 2 are never stored: a
                        //  compile-time test per unrolled s, so the butterfly's unused results are not even computed)
                        if (C::HS && s * P > C::N / 2) continue;
                        V y = v[k][s];
                        if (p.inverse) {
                            y.x *= (T)p.scale;
                            y.y *= -(T)p.scale;
                        }
                        if (!C::HS || o0 + s * P <= p.store_lim) gstore<(C::NT & 2) != 0>(gout + base + s * step + off, y);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < R; ++s) lds[lds_index<C, I>(c, o0 + s * P)] = v[k][s];
            }
        }
        MIFFT_FENCE();
    }
}

// A pass with a prime radix R > 32 (C::BIGP(I)), cooperatively in LDS.  The tile is in LDS (pass 0: staged by the copy).
//   1. in place: a_j = x_j + x_{R-j} at position j, b_j = x_j - x_{R-j} at position R-j   (j = 1..H, H = (R-1)/2)
//   2. per output pair s = 0..H (four consecutive s per thread):  A = x_0 + sum_j cos(2 pi j s / R) a_j,  B = sum_j sin(2 pi j s / R) b_j
//      X_s = A - iB, X_{R-s} = A + iB  -- the conjugate-pair form of DftOddPrime, 4 real FMAs per (j, pair) instead of
//      the 8 of the literal stage (fft/fft/_fft.mojo:261-290); lanes that share a butterfly read a_j, b_j as broadcasts
//   3. after a barrier the outputs go to their Stockham positions
//   (pass I >= 1: the inputs x_j are first multiplied by their Stockham twiddles W_{P R}^{j p} from the LDS table, and
//    the outputs go to q*P*R + p + s*P)
template <class C, int I>
MIFFT_DEV void bigprime_pass(cpx<typename C::T>* lds, const cpx<typename C::T>* ltw, const cpx<typename C::T>* cs, int tid) {
    using T = typename C::T;
    using V = cpx<T>;
    constexpr int R = C::R(I), H = (R - 1) / 2, NB = C::NB(I), P = C::P(I);
    constexpr int PAIRS = C::TILE * NB * H;
    for (int e = tid; e < PAIRS; e += C::THREADS) {
        const int c = e / (NB * H), rem = e - c * (NB * H);
        const int b = rem / H, j = rem - b * H + 1;
        const int i1 = lds_index<C, I - 1>(c, b + j * NB), i2 = lds_index<C, I - 1>(c, b + (R - j) * NB);
        V u = lds[i1], v = lds[i2];
        if constexpr (I > 0) {
            const int pp = b % P;
            u = cmul(u, ltw[C::TWL_OFF(I) + (j - 1) * P + pp]);
            v = cmul(v, ltw[C::TWL_OFF(I) + (R - j - 1) * P + pp]);
        }
        lds[i1] = u + v;
        lds[i2] = u - v;
    }
    __syncthreads();
    // item = (transform, butterfly, group of SB consecutive s): a_j and b_j are read once per j for SB output pairs
    constexpr int SB = MIFFT_BIGP_SB, GROUPS = (H + 1 + SB - 1) / SB;
    constexpr int ITEMS = C::TILE * NB * GROUPS;
    constexpr int IPT = (ITEMS + C::THREADS - 1) / C::THREADS;
    V lo[IPT][SB], hi[IPT][SB];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int id = tid + k * C::THREADS;
        if (id < ITEMS) {
            const int cb = id / GROUPS, s0 = (id - cb * GROUPS) * SB;
            const int c = cb / NB, b = cb - c * NB;
            const V x0 = lds[lds_index<C, I - 1>(c, b)];
            V A[SB], B[SB];
            int m[SB];
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                A[q] = x0;
                B[q] = {(T)0, (T)0};
                m[q] = 0;
            }
#pragma unroll 2
            for (int j = 1; j <= H; ++j) {
                const V a = lds[lds_index<C, I - 1>(c, b + j * NB)], d = lds[lds_index<C, I - 1>(c, b + (R - j) * NB)];
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    // s0 + q may run past H in the last group: its index stays in range (s <= H + SB - 1 < R) and
                    // its result is never stored
                    m[q] += s0 + q;
                    if (m[q] >= R) m[q] -= R;
                    const V w = cs[m[q]];  // (cos, sin)(2 pi m / R)
                    A[q].x = fma_t(w.x, a.x, A[q].x);
                    A[q].y = fma_t(w.x, a.y, A[q].y);
                    B[q].x = fma_t(w.y, d.x, B[q].x);
                    B[q].y = fma_t(w.y, d.y, B[q].y);
                }
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                lo[k][q] = {A[q].x + B[q].y, A[q].y - B[q].x};  // X_s
                hi[k][q] = {A[q].x - B[q].y, A[q].y + B[q].x};  // X_{R-s}
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int id = tid + k * C::THREADS;
        if (id < ITEMS) {
            const int cb = id / GROUPS, s0 = (id - cb * GROUPS) * SB;
            const int c = cb / NB, b = cb - c * NB;
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const int s = s0 + q;
                if (s <= H) {
                    const int qq = b / P, o0 = qq * P * R + (b - qq * P);  // Stockham scatter (pass 0: P = 1)
                    lds[lds_index<C, I>(c, o0 + s * P)] = lo[k][q];
                    if (s > 0) lds[lds_index<C, I>(c, o0 + (R - s) * P)] = hi[k][q];
                }
            }
        }
    }
    __syncthreads();
}

// One sub-pass of the M-point FFT inside rader_pass.  K: pass of the split; INV: second (inverse) transform.
//   gather    K = 0 of the forward transform reads the INPUT positions of the R-point DFT through the permutation g^q
//             (times the Stockham twiddle of the enclosing pass when I > 0); every other sub-pass reads slots 1..M of the
//             DFT's OUTPUT block
//   finish    last sub-pass forward:  slot 0 <- x_0 + A[0];  A[n] <- conj(A[n] * B[n])   (B carries the 1 / M)
//             last sub-pass inverse:  X[g^-q] = x_0 + conj(d[q]) scattered through the second permutation
template <class C, int I, int K, bool INV>
MIFFT_DEV void rader_sub(cpx<typename C::T>* lds, const cpx<typename C::T>* ltw, const cpx<typename C::T>* rt, int tid) {
    using T = typename C::T;
    using V = cpx<T>;
    constexpr int R = C::R(I), M = R - 1, NB = C::NB(I), P = C::P(I), INST = C::TILE * NB;
    constexpr bool PAD = !rader_ok(R);   // convolution of length L > M in a scratch block of L elements per DFT
    constexpr int L = rader_len(R);
    constexpr RaderSplitT S = rader_split(L);
    constexpr int RK = S.r[K], NBK = L / RK;
    constexpr int PK = (K == 0 ? 1 : S.r[0]) * (K <= 1 ? 1 : S.r[1]) * (K <= 2 ? 1 : S.r[2]);  // product of the earlier radices
    constexpr int RATIO = L / (PK * RK);
    constexpr int ITEMS = INST * NBK, IPT = (ITEMS + C::THREADS - 1) / C::THREADS;
    constexpr bool LAST = K == S.np - 1;
    const V* Bt = rt;
    const V* Wm = rt + L;
    const unsigned short* perm_in = (const unsigned short*)(rt + 2 * L);
    const unsigned short* perm_out = perm_in + M;
    V* x0buf = (V*)rt + 2 * L + (4 * M + (int)sizeof(V) - 1) / (int)sizeof(V);
    V* scr = x0buf + INST;  // PAD only: [inst][L]
    // element n of the running convolution of DFT `inst` sits at blk[n * STR]: slot 1 + n of its output block in the data
    // tile (the Stockham layout is linear in n: rows stride P, column tiles P * CPITCH), or its scratch block
    constexpr int STR = PAD ? 1 : (C::COLS ? P * C::CPITCH : P);
    static_assert(PAD || C::COLS || !is_pow2_ce(C::N), "no swizzle on lengths with a prime factor above 32");
    V v[IPT][RK];
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
        const int id = tid + it * C::THREADS;
        if (ITEMS % C::THREADS == 0 || id < ITEMS) {
            const int inst = id / NBK, kb = id - inst * NBK;
            const int c = inst / NB, b = inst - c * NB;
            const int q = b / P, pp = b - q * P, o0 = q * P * R + pp;
            if constexpr (K == 0 && !INV) {
#pragma unroll
                for (int t = 0; t < RK; ++t) {
                    const int qq = kb + t * NBK;
                    V u = {(T)0, (T)0};
                    if (!PAD || qq < M) {  // (padded: the sequence x[g^q] is followed by L - M zeros)
                        const int j = perm_in[qq];  // 1 .. R-1
                        u = lds[lds_index<C, I - 1>(c, b + j * NB)];
                        if constexpr (I > 0) u = cmul(u, ltw[C::TWL_OFF(I) + (j - 1) * P + pp]);
                    }
                    v[it][t] = u;
                }
                if (kb == 0) x0buf[inst] = lds[lds_index<C, I - 1>(c, b)];  // (x0buf is outside the data tile)
            } else {
                const int ppk = kb % PK;
                const V* blk = PAD ? scr + inst * L : lds + lds_index<C, I>(c, o0 + P);
#pragma unroll
                for (int t = 0; t < RK; ++t) {
                    V u = blk[(kb + t * NBK) * STR];
                    if (t > 0 && PK > 1) u = cmul(u, Wm[t * ppk * RATIO]);
                    v[it][t] = u;
                }
            }
        }
    }
    __syncthreads();  // every read of this sub-pass precedes every write (the blocks of different DFTs interleave)
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
        const int id = tid + it * C::THREADS;
        if (ITEMS % C::THREADS == 0 || id < ITEMS) {
            const int inst = id / NBK, kb = id - inst * NBK;
            const int c = inst / NB, b = inst - c * NB;
            const int q = b / P, pp = b - q * P, o0 = q * P * R + pp;
            Dft<RK, T, 1>::run(v[it]);
            const int qk = kb / PK, ppk = kb - qk * PK, n0 = qk * PK * RK + ppk;  // outputs n0 + s * PK
            V* blk = PAD ? scr + inst * L : lds + lds_index<C, I>(c, o0 + P);
#pragma unroll
            for (int s2 = 0; s2 < RK; ++s2) {
                const int n = n0 + s2 * PK;
                V y = v[it][s2];
                if constexpr (LAST && !INV) {
                    if (n == 0) lds[lds_index<C, I>(c, o0)] = x0buf[inst] + y;  // X_0 = x_0 + sum of the others
                    y = cmul(y, Bt[n]);
                    y.y = -y.y;
                    blk[n * STR] = y;
                } else if constexpr (LAST && INV) {
                    y.y = -y.y;
                    if (!PAD || n < M) lds[lds_index<C, I>(c, o0 + (int)perm_out[n] * P)] = x0buf[inst] + y;
                } else {
                    blk[n * STR] = y;
                }
            }
        }
    }
    __syncthreads();
    if constexpr (!LAST) rader_sub<C, I, K + 1, INV>(lds, ltw, rt, tid);
}

// A prime radix R > 32 (C::RADER(I)): Rader's algorithm inside the LDS tile; replaces bigprime_pass.  In place in the DFT's
// output block when R - 1 splits into register butterflies, through a zero-padded convolution in a scratch block otherwise.
// Inputs at positions b + j NB of the staged / previous layout, outputs at the Stockham positions q P R + p + s P.
template <class C, int I>
MIFFT_DEV void rader_pass(cpx<typename C::T>* lds, const cpx<typename C::T>* ltw, const cpx<typename C::T>* rt, int tid) {
    rader_sub<C, I, 0, false>(lds, ltw, rt, tid);  // A = FFT_M(x[g^q]);  slot 0 <- X_0;  slots 1.. <- conj(A B)
    rader_sub<C, I, 0, true>(lds, ltw, rt, tid);   // d = FFT_M(.);  X[g^-q] = x_0 + conj(d[q])
}

// 1: the next tile's HBM loads are issued in slices between the passes of the current tile; 0: all at the top
#ifndef MIFFT_SLICED_PREFETCH
#define MIFFT_SLICED_PREFETCH 1
#endif
#ifndef MIFFT_SLICED_PREFETCH_PLANE
#define MIFFT_SLICED_PREFETCH_PLANE 1
#endif

template <int K>
struct IntC {
    static constexpr int value = K;
};

struct NoHook {
    MIFFT_DEV void operator()() const {}
    template <int K>
    MIFFT_DEV void operator()(IntC<K>) const {}
};

// TWSHIFT: extra offset of this configuration's LDS twiddle table (rectangular planes keep two tables)
// `before_stores` runs once per tile, after the last LDS read and right BEFORE the butterflies + HBM stores of the last
// pass: the prefetching kernels wait there for the next tile's inputs (loads issued a whole tile earlier: the wait is
// free).  vmcnt counts loads and stores together and the compiler's s_waitcnt insertion cannot count the exec-masked
// stores, so without this the wait for the prefetched registers -- at the top of the next iteration -- was a vmcnt(0):
// it waited for every store of the tile just finished before the next loads were even issued (29 % of the tile period
// of the 640-point column tile by in-kernel stamps).  After an explicit, compiler-visible vmcnt(0) ahead of the stores
// the registers are known to be loaded and the top-of-loop copy needs no wait at all.
// `between(IntC<K>)` runs at the seams of the tile: K = 2 I + 1 behind the barrier that follows the scatter of pass I,
// K = 2 I behind the barrier that follows the gather of pass I (1 <= I < NP - 1).  The prefetching kernels issue the
// next tile's HBM loads there in SLICES.  Issued in one burst (at the top of the tile, or anywhere else) the 20 loads
// per thread of a 640-point column tile take ~5 000 cycles to be ACCEPTED -- every wave of the one workgroup a CU holds
// sits in that issue stall at once, and nothing computes meanwhile (in-kernel stamps: the stall moves with the burst,
// 4 500 of 16 600 cycles per tile).  A few loads per seam keep the vector-memory queue short, the waves go on to their
// butterflies, and the memory system sees a steady stream instead of load / compute / store phases.
template <class C, int I, int TWSHIFT = 0, class Hook = NoHook, class Hook0 = NoHook>
MIFFT_DEV void run_pass(const TileParams& p, cpx<typename C::T>* lds, const cpx<typename C::T>* twr,
                        cpx<typename C::T> (*pre)[C::R(0)], long long base, int nv, int tid, long long obase = 0,
                        int fs_row = 0, Hook before_stores = Hook(), Hook0 between = Hook0()) {
    if constexpr (I < C::NP && C::RADER(I)) {
        rader_pass<C, I>(lds, lds + C::DATA_ELEMS + TWSHIFT, lds + C::DATA_ELEMS + C::TWL_TOTAL + C::CS_OFF(I), tid);
        run_pass<C, I + 1, TWSHIFT>(p, lds, twr, pre, base, nv, tid, obase, fs_row, before_stores, between);
    } else if constexpr (I < C::NP && C::BIGP(I)) {
        bigprime_pass<C, I>(lds, lds + C::DATA_ELEMS + TWSHIFT, lds + C::DATA_ELEMS + C::TWL_TOTAL + C::CS_OFF(I), tid);
        run_pass<C, I + 1, TWSHIFT>(p, lds, twr, pre, base, nv, tid, obase, fs_row, before_stores, between);
    } else if constexpr (I < C::NP) {
        using T = typename C::T;
        using V = cpx<T>;
        constexpr int R = C::R(I), IPT = C::IPT(I);
        constexpr bool SRC_GLOBAL = (I == 0) && C::FIRST_DIRECT;
        constexpr bool DST_GLOBAL = (I == C::NP - 1) && C::LAST_DIRECT;
        V v[IPT][R];
        if constexpr (SRC_GLOBAL) {
            if constexpr (C::PREFETCH) {
#pragma unroll
                for (int k = 0; k < IPT; ++k)
#pragma unroll
                    for (int j = 0; j < R; ++j) v[k][j] = pre[k][j];
            } else {
                load_pass0<C>(p, v, base, nv, tid);
            }
            // inverse = conj(F(conj x)) / N.  A real input is its own conjugate: its imaginary parts stay the CONSTANT +0,
            // which lets the compiler fold the imaginary half of the first butterflies away (radix 31 on real input: 176 ->
            // VGPRs, no spill; a runtime -0 / +0 kept every one of them live).
            if (p.inverse && C::CONJ_IN) {
#pragma unroll
                for (int k = 0; k < IPT; ++k)
#pragma unroll
                    for (int j = 0; j < R; ++j) v[k][j].y = -v[k][j].y;
            }
        } else {
            pass_gather_lds<C, I>(p, lds, lds + C::DATA_ELEMS + TWSHIFT, twr, v, tid);
            // in-place LDS buffer: every read of this pass completes before any later write (this
            // pass's scatter, or pass 0 of the NEXT tile when this pass stores to HBM).  WSUB: the passes 1..NP-2
            // exchange inside wave-owned sub-problems -- only the LAST gather must hold back the other waves' next
            // tile
            if constexpr (C::WSUB && I >= 1 && I < C::NP - 1)
                wave_lds_fence();
            else
                wg_barrier<C>();
            if constexpr (I < C::NP - 1) between(IntC<2 * I>{});
        }
        MIFFT_STAMP(C, 1 + 4 * I);  // gather (+ its barrier) of pass I: LDS reads / twiddles / wait for HBM loads
        if constexpr (I == C::NP - 1) before_stores();
        pass_compute_scatter<C, I>(p, lds, v, base, nv, tid, obase, fs_row);
        MIFFT_STAMP(C, 2 + 4 * I);  // butterflies + scatter issue of pass I
        if constexpr (!DST_GLOBAL) {
            if constexpr (C::WSUB && I >= 1 && I < C::NP - 1)
                wave_lds_fence();
            else
                wg_barrier<C>();
        }
        MIFFT_STAMP(C, 3 + 4 * I);  // barrier after the scatter of pass I
        if constexpr (!DST_GLOBAL && I < C::NP - 1) between(IntC<2 * I + 1>{});
        run_pass<C, I + 1, TWSHIFT>(p, lds, twr, pre, base, nv, tid, obase, fs_row, before_stores, between);
    }
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void tile_kernel(const TileParams p) {
    using T = typename C::T;
    using V = cpx<T>;
#ifdef MIFFT_STATIC_LDS  // runtime-compiled instances (kernels_jit.cpp): no per-function dynamic-LDS opt-in needed
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
    V* lds = (V*)smem;
    const int tid0 = threadIdx.x;
    V twr[(C::TWMODE == TW_REG && C::TW_TOTAL > 0) ? C::TW_TOTAL : 1];
    if constexpr (C::TWMODE == TW_REG) preload_tw<C, 1>(twr, (const V*)p.tw, tid0, p.inverse);
    if constexpr (C::TWMODE == TW_LDS) {
        fill_lds_tw<C, 1>(lds + C::DATA_ELEMS, (const V*)p.tw, tid0, p.inverse);
        // cooperative passes: (cos, sin)(2 pi m / R) from W_N^(m N/R) = cos - i sin; Rader passes: the plan's tables
        // (p.tlo: spectrum of the kernel and W_M per Rader pass, complex; p.thi: the two permutations per Rader pass, uint16)
        if constexpr (C::BIGP0 && !C::RADER(0)) {
            for (int m = tid0; m < C::R(0); m += C::THREADS) {
                V w = ((const V*)p.tw)[m * C::NB(0)];
                if (p.inverse) w.y = -w.y;
                lds[C::DATA_ELEMS + C::TWL_TOTAL + m] = {w.x, -w.y};
            }
        }
        if constexpr (C::BIGP1 && !C::RADER(1)) {
            for (int m = tid0; m < C::R(1); m += C::THREADS) {
                V w = ((const V*)p.tw)[m * C::NB(1)];
                if (p.inverse) w.y = -w.y;
                lds[C::DATA_ELEMS + C::TWL_TOTAL + C::CS_OFF(1) + m] = {w.x, -w.y};
            }
        }
        if constexpr (C::RADER(0) || C::RADER(1)) {
            // per Rader pass: complex [B (L) | W_L (L)] from p.tlo, uint16 [g^q (M) | g^-q (M)] from p.thi.  Everything here
            // is a compile-time constant (a runtime call of rader_len would factorise thousands of integers per thread)
            constexpr int L0 = C::RADER(0) ? rader_len(C::R(0)) : 0, M0 = C::RADER(0) ? C::R(0) - 1 : 0;
            constexpr int L1 = C::RADER(1) ? rader_len(C::R(1)) : 0, M1 = C::RADER(1) ? C::R(1) - 1 : 0;
            if constexpr (C::RADER(0)) {
                V* dst = lds + C::DATA_ELEMS + C::TWL_TOTAL + C::CS_OFF(0);
                for (int m = tid0; m < 2 * L0; m += C::THREADS) dst[m] = ((const V*)p.tlo)[m];
                unsigned short* pd = (unsigned short*)(dst + 2 * L0);
                for (int m = tid0; m < 2 * M0; m += C::THREADS) pd[m] = ((const unsigned short*)p.thi)[m];
            }
            if constexpr (C::RADER(1)) {
                V* dst = lds + C::DATA_ELEMS + C::TWL_TOTAL + C::CS_OFF(1);
                for (int m = tid0; m < 2 * L1; m += C::THREADS) dst[m] = ((const V*)p.tlo)[2 * L0 + m];
                unsigned short* pd = (unsigned short*)(dst + 2 * L1);
                for (int m = tid0; m < 2 * M1; m += C::THREADS) pd[m] = ((const unsigned short*)p.thi)[2 * M0 + m];
            }
        }
        __syncthreads();
    }

    MIFFT_STAMP(C, -1);
    V pre[C::PREFETCH ? C::IPT(0) : 1][C::R(0)];
    // this workgroup's tiles: t, t + t_step, ... < t_end
    long long t = blockIdx.x, t_end = p.n_tiles, t_step = gridDim.x;
    long long run_begin = 0;
    if constexpr (C::HERM_RUNS) {  // one contiguous run of tiles per workgroup (lengths differ by one at most), walked in
                                   // descending column order (HERM stores)
        const long long len = p.n_tiles / gridDim.x, rem = p.n_tiles - len * gridDim.x, w = blockIdx.x;
        run_begin = t = w * len + (w < rem ? w : rem);
        t_end = t + len + (w < rem ? 1 : 0);
        t_step = 1;
    }
    // (runs: position t of the order is tile n_tiles - 1 - t, whatever p.reverse says)
    auto tile_at = [&](long long pos) { return C::HERM_RUNS ? p.n_tiles - 1 - pos : tile_id(p, pos); };
    if constexpr (C::XCD_CHUNK) {
        if (gridDim.x >= 8) {  // (smaller grids: some XCD would own tiles but no workgroup)
            const long long x = blockIdx.x & 7, slot = blockIdx.x >> 3;
            t_step = ((long long)gridDim.x - x + 7) >> 3;  // workgroups on this XCD
            t = x * p.n_tiles / 8 + slot;
            t_end = (x + 1) * p.n_tiles / 8;
        }
    }
    static_assert(!(C::PREFETCH && C::BIGP0), "prefetching kernels read pass 0 straight from HBM");
    if constexpr (C::PREFETCH) {
        if (t < t_end) {
            long long base;
            int nv;
            if constexpr (C::HERM) {
                int col0, in_row;
                tile_geom_herm<C>(p, tile_at(t), base, nv, col0, in_row);
            } else {
                tile_geom<C>(p, tile_at(t), base, nv);
            }
            load_pass0<C>(p, pre, base, nv, tid0);
        }
    }
    const int tid_entry = tid0;
    for (; t < t_end; t += t_step) {
        // The thread index is made opaque once per tile: every LDS / twiddle / HBM offset derived from it is then
        // recomputed inside the iteration (a few dozen VALU operations) instead of being hoisted out of the
        // persistent loop, where dozens of loop-invariant address registers stay live across the whole tile and
        // push the kernel over its VGPR budget.
        int tid = tid_entry;
#ifndef MIFFT_NO_OPAQUE_TID
        if constexpr (C::OPAQUE_TID) asm volatile("" : "+v"(tid));
#endif
        long long base;
        int nv;
        const long long tt = tile_at(t);
        long long obase = 0;
        int fs_row = 0;
        if constexpr (C::FS1) {
            tile_geom_fs1<C>(p, tt, base, obase, nv, fs_row);
        } else if constexpr (C::HERM) {  // image base and first column of the tile, for the mirrored stores
            int in_row;
            tile_geom_herm<C>(p, tt, base, nv, fs_row, in_row);
            obase = base - fs_row;
            // tile tt hands its first column to tile tt - 1 when that is the next of this run, covers the columns right below
            // it, and the column has a mirror image to be stored at all; the same test one tile up says whether tile tt + 1
            // did so (not across the start of a row of the last dimension: that column's mirror image is the FIRST element of
            // a row, next to nothing the neighbouring tile stores)
            if constexpr (C::HERM_RUNS) {
                const int nxt = fs_row + C::TILE;
                const bool out = t + 1 < t_end && fs_row % p.herm_d2 != 0 && herm_twice(p, fs_row);
                const bool in = t > run_begin && in_row + 1 < p.herm_tpr && nxt % p.herm_d2 != 0 && herm_twice(p, nxt);
                fs_row |= (in ? 1 << 30 : 0) | (out ? (int)(1u << 31) : 0);
            }
        } else {
            tile_geom<C>(p, tt, base, nv);
        }
        V cur[C::PREFETCH ? C::IPT(0) : 1][C::R(0)];
        // seams between passes where a slice of the next tile's loads can go: 2 NP - 3 of them, plus the top
        // (column tiles only: row tiles measured 4-8 % SLOWER sliced -- rows480 0.096 -> 0.104 ms, tools/tune GROUP 5 --
        //  while 640- and 1024-point column tiles gain 4-6 %)
        constexpr int SLICES = (C::PREFETCH && C::COLS && MIFFT_SLICED_PREFETCH && C::NP > 1) ? 2 * C::NP - 2 : 1;
        const long long tn = t + t_step;
        long long nbase = 0;
        int nnv = 0;
        if constexpr (C::PREFETCH) {
#pragma unroll
            for (int k = 0; k < C::IPT(0); ++k)
#pragma unroll
                for (int j = 0; j < C::R(0); ++j) cur[k][j] = pre[k][j];
            if (tn < t_end) {
                if constexpr (C::HERM) {
                    int col0, in_row;
                    tile_geom_herm<C>(p, tile_at(tn), nbase, nnv, col0, in_row);
                } else {
                    tile_geom<C>(p, tile_at(tn), nbase, nnv);
                }
            }
            if constexpr (SLICES == 1) {
                if (tn < t_end) load_pass0<C>(p, pre, nbase, nnv, tid);  // all of the next tile's HBM reads at once
            } else {
                if (tn < t_end) load_pass0<C, 0, SLICES>(p, pre, nbase, nnv, tid);
            }
        }
        // the other slices of the next tile's loads, one per seam between the passes (see run_pass)
        auto slice = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (C::PREFETCH && SLICES > 1 && K >= 1 && K < SLICES) {
                if (tn < t_end) load_pass0<C, K, SLICES>(p, pre, nbase, nnv, tid);
            }
        };
        // Run by the last pass right before its stores (see run_pass): the next tile's inputs were issued during this
        // tile, so waiting for them HERE costs little -- and it keeps the compiler's wait for them (at the copy above,
        // next iteration) from landing behind this tile's stores.
        auto drain = [&]() {
            if constexpr (C::PREFETCH) vm_drain();
        };
        if constexpr (!C::FIRST_DIRECT && C::COLS) {
            // column tile staged in LDS (only the big-prime pass 0 needs this): runs of TILE adjacent columns
            static_assert(C::BIGP0, "column tiles load directly unless pass 0 works in LDS");
            for (int f = tid; f < C::N * C::TILE; f += C::THREADS) {
                const int n = f / C::TILE, c = f - n * C::TILE;
                const int cc = c < nv ? c : nv - 1;  // ragged last tile: re-read a valid column, never stored
                V x;
                if constexpr (!same_t<typename C::IT, T>::value)
                    x = load_foreign<C>(p.in, gaddr<C>(p, base, cc, n));
                else if constexpr (C::IN_REAL)
                    x = {gload_real<false>((const T*)p.in + gaddr<C>(p, base, cc, n)), (T)0};
                else
                    x = gload<false>((const V*)p.in + gaddr<C>(p, base, cc, n));
                if (p.inverse && C::CONJ_IN) x.y = -x.y;
                lds[lds_index<C, -1>(c, n)] = x;
            }
            __syncthreads();
        } else if constexpr (!C::FIRST_DIRECT) {
            // flat, fully coalesced HBM -> LDS copy of the tile (rows need not be 16-B aligned: N = 93)
            const V* gin = (const V*)p.in;
            const int total = nv * C::N;
            {
                for (int f = tid; f < total; f += C::THREADS) {
                    const int c = f / C::N, n = f - c * C::N;
                    V x = {(T)0, (T)0};
                    if constexpr (!same_t<typename C::IT, T>::value)
                        x = load_foreign<C>(p.in, base + f);
                    else if constexpr (C::IN_REAL)
                        x.x = gload_real<(C::NT & 1) != 0>((const T*)p.in + base + f);
                    else
                        x = gload<(C::NT & 1) != 0>(gin + base + f);
                    if (p.inverse && C::CONJ_IN) x.y = -x.y;
                    lds[lds_index<C, -1>(c, n)] = x;
                }
            }
            __syncthreads();
        }
        MIFFT_STAMP(C, 0);  // tile bookkeeping + issue of the next tile's prefetch
        run_pass<C, 0>(p, lds, twr, cur, base, nv, tid, obase, fs_row, drain, slice);
        if constexpr (C::TSTORE) {
            // transposed + twiddled flat store: out[o][c0 + c][k1] = tile[k1][c] * W^{k1 * (c0 + c)}
            const long long o = tt / p.tiles_per_outer;
            const long long c0 = (tt - o * p.tiles_per_outer) * C::TILE;
            V* gout = (V*)p.out + o * (long long)C::N * p.inner + c0 * (long long)C::N;
            const V* tlo = (const V*)p.tlo;
            const V* thi = (const V*)p.thi;
            if constexpr (C::N % C::THREADS == 0 || C::THREADS % C::N == 0) {
                // a thread keeps its k1 for the whole tile: W^(k1 * (c0 + c)) = W^(k1 * c0) [two-level lookup, once
                // per k1] * W^(k1 * c) [plan-time table, read contiguously along k1]
                constexpr int KPT = C::N >= C::THREADS ? C::N / C::THREADS : 1;
                constexpr int CSTEP = C::N >= C::THREADS ? 1 : C::THREADS / C::N;
                const int k1b = C::N >= C::THREADS ? tid : tid % C::N;
                const int cb = C::N >= C::THREADS ? 0 : tid / C::N;
                const V* tcol = (const V*)p.tcol;
                V a[KPT];
#pragma unroll
                for (int kk = 0; kk < KPT; ++kk) {
                    const long long m0 = (long long)(k1b + kk * C::THREADS) * c0;  // < N * inner
                    a[kk] = cmul(tlo[m0 & 1023], thi[m0 >> 10]);
                }
                // all table reads of the tile are issued before the first use (the table is L2-resident)
                constexpr int CIT = (C::TILE + CSTEP - 1) / CSTEP;
                V w[CIT][KPT];
                if constexpr (CSTEP == 1 && C::TILE % 4 == 0) {
                    // only rows 1, 2, 3 and 4, 8, 12, ... of the table are read:
                    // W^(k1 * (c0 + 4 q + r)) = [W^(k1 * c0) * W^(k1 * 4 q)] * W^(k1 * r)
                    constexpr int Q = C::TILE / 4;
                    V lo[3][KPT], hi[Q][KPT];
#pragma unroll
                    for (int kk = 0; kk < KPT; ++kk) {
                        const int k1 = k1b + kk * C::THREADS;
#pragma unroll
                        for (int r = 1; r < 4; ++r) lo[r - 1][kk] = tcol[r * C::N + k1];
#pragma unroll
                        for (int q = 1; q < Q; ++q) hi[q][kk] = tcol[4 * q * C::N + k1];
                    }
#pragma unroll
                    for (int kk = 0; kk < KPT; ++kk) {
                        hi[0][kk] = a[kk];
#pragma unroll
                        for (int q = 1; q < Q; ++q) hi[q][kk] = cmul(a[kk], hi[q][kk]);
#pragma unroll
                        for (int i = 0; i < CIT; ++i)
                            w[i][kk] = (i % 4 == 0) ? hi[i / 4][kk] : cmul(hi[i / 4][kk], lo[i % 4 - 1][kk]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < CIT; ++i) {
                        const int c = cb + i * CSTEP;
                        const int cc = c < C::TILE ? c : C::TILE - 1;
#pragma unroll
                        for (int kk = 0; kk < KPT; ++kk) w[i][kk] = cmul(a[kk], tcol[cc * C::N + k1b + kk * C::THREADS]);
                    }
                }
#pragma unroll
                for (int i = 0; i < CIT; ++i) {
                    const int c = cb + i * CSTEP;
                    if (c < nv) {
#pragma unroll
                        for (int kk = 0; kk < KPT; ++kk) {
                            const int k1 = k1b + kk * C::THREADS;
                            V y = cmul(lds[k1 * C::CPITCH + c], w[i][kk]);
                            if (p.inverse) {  // conj(F(conj x)) * conj(W) * 1/N = conj(F(conj x) * W) * 1/N
                                y.x *= (T)p.scale;
                                y.y *= -(T)p.scale;
                            }
                            gstore<(C::NT & 2) != 0>(gout + c * C::N + k1, y);
                        }
                    }
                }
            } else {
                const int total = nv * C::N;
                for (int f = tid; f < total; f += C::THREADS) {
                    const int c = f / C::N, k1 = f - c * C::N;
                    const long long m = (long long)k1 * (c0 + c);  // < N * inner
                    V y = cmul(lds[k1 * C::CPITCH + c], cmul(tlo[m & 1023], thi[m >> 10]));
                    if (p.inverse) {
                        y.x *= (T)p.scale;
                        y.y *= -(T)p.scale;
                    }
                    gstore<(C::NT & 2) != 0>(gout + f, y);
                }
            }
            __syncthreads();
        } else if constexpr (!C::LAST_DIRECT && C::COLS) {
            // a strided PRIME length (one cooperative pass, result in LDS): runs of TILE adjacent columns
            static_assert(C::BIGP(C::NP - 1), "column tiles store directly unless the last pass works in LDS");
            V* gout = (V*)p.out;
            for (int f = tid; f < C::N * C::TILE; f += C::THREADS) {
                const int n = f / C::TILE, c = f - n * C::TILE;
                if (c < nv && (!C::HS || n <= p.store_lim)) {
                    V y = lds[lds_index<C, C::NP - 1>(c, n)];
                    if (p.inverse) {
                        y.x *= (T)p.scale;
                        y.y *= -(T)p.scale;
                    }
                    gout[gaddr<C>(p, base, c, n)] = y;
                }
            }
            __syncthreads();
        } else if constexpr (C::R2C) {
            // Z = F(z) of the packed row lies in LDS.  E[k] = (Z[k] + conj Z[N-k]) / 2 and O[k] = (Z[k] - conj Z[N-k]) / 2i are
            // the transforms of the even and the odd samples; X[k] = E[k] + W^k O[k] and X[N-k] = conj(E[k] - W^k O[k]) with
            // W = e^(-2 pi i / 2N); k = 0 gives X[0] and X[N].  One thread per pair (k, N - k), both runs of stores contiguous.
            V* gout = (V*)p.out;
            const V* w = (const V*)p.r2c_tw;
            constexpr int PAIRS = C::N / 2 + 1;
            const long long row0 = base / C::N;
            for (int f = tid; f < nv * PAIRS; f += C::THREADS) {
                const int c = f / PAIRS, k = f - c * PAIRS, m = k ? C::N - k : 0;
                const V zk = lds[lds_index<C, C::NP - 1>(c, k)], zm = lds[lds_index<C, C::NP - 1>(c, m)];
                const T ex = (T)0.5 * (zk.x + zm.x), ey = (T)0.5 * (zk.y - zm.y);
                const T ox = (T)0.5 * (zk.y + zm.y), oy = (T)-0.5 * (zk.x - zm.x);
                const V wk = w[k];
                const T tx = wk.x * ox - wk.y * oy, ty = wk.x * oy + wk.y * ox;
                V a = {ex + tx, ey + ty}, b = {ex - tx, ty - ey};
                if (p.inverse) {  // the inverse of a real row: conj(X) / 2N
                    a.x *= (T)p.scale;
                    a.y *= -(T)p.scale;
                    b.x *= (T)p.scale;
                    b.y *= -(T)p.scale;
                }
                V* row = gout + (row0 + c) * p.out_pitch;
                gstore<(C::NT & 2) != 0>(row + k, a);
                if (2 * k != C::N) gstore<(C::NT & 2) != 0>(row + C::N - k, b);
            }
            __syncthreads();
        } else if constexpr (!C::LAST_DIRECT) {
            V* gout = (V*)p.out;
            const int total = nv * C::N;
            for (int f = tid; f < total; f += C::THREADS) {
                const int c = f / C::N, n = f - c * C::N;
                if (C::HS && n > p.store_lim) continue;
                V y = lds[lds_index<C, C::NP - 1>(c, n)];
                if (p.inverse) {
                    y.x *= (T)p.scale;
                    y.y *= -(T)p.scale;
                }
                gstore<(C::NT & 2) != 0>(gout + base + f, y);
            }
            __syncthreads();
        }
    }
    MIFFT_STAMP_DUMP(C, p);
}


// ---------------------------------------------------------------------------------------------
// plane_kernel<CR, CC>: the two innermost dimensions (N1 x N2, N2 contiguous) of one 2-D slice are
// transformed inside ONE LDS tile: the rows (length N2) by the ROWS configuration CR -- pass 0 straight
// from HBM, last pass left in LDS in natural order -- then the columns (length N1, LDS stride N2) by
// the COLS configuration CC, whose last pass stores straight to HBM along the contiguous axis.
// One HBM read + one HBM write for two dimensions; the reference spends a row kernel, a transpose
// kernel, a row kernel and a transpose back (fft/fft/_ndim_fft_gpu.mojo:634-642).
// Requirements: CR::N == CC::TILE (= N2), CR::TILE == CC::N (= N1), CR::LD == N2, equal thread counts,
// identical LDS twiddle tables (N1 == N2 with the same radices).
// ---------------------------------------------------------------------------------------------
template <class CR, class CC>
__global__ __launch_bounds__(CR::THREADS, CR::MINW) void plane_kernel(const TileParams p) {
    using T = typename CR::T;
    using V = cpx<T>;
    static_assert(!CR::COLS && CC::COLS, "rows configuration first, columns configuration second");
    static_assert(CR::N == CC::TILE && CR::TILE == CC::N && CR::LD == CR::N, "plane geometry");
    static_assert(CR::THREADS == CC::THREADS, "one thread count");
    static_assert(CR::FIRST_DIRECT && !CR::LAST_DIRECT && !CC::FIRST_DIRECT && CC::LAST_DIRECT, "plane data flow");
    static_assert(CR::TWMODE == TW_LDS && CC::TWMODE == TW_LDS, "planes keep their twiddles in LDS");
    // square planes with the same radices share one LDS twiddle table; rectangular planes keep the column table (W_N1,
    // from p.tlo) right behind the row table (W_N2, from p.tw)
    constexpr bool SHARED_TW = CR::N == CC::N && CR::NP == CC::NP && CR::R(0) == CC::R(0) && CR::R(1) == CC::R(1) &&
                               CR::R(2) == CC::R(2) && CR::R(3) == CC::R(3);
    constexpr int CSHIFT = SHARED_TW ? 0 : CR::TWL_TOTAL;
    constexpr size_t PLANE_LDS = (size_t)(CR::DATA_ELEMS + CSHIFT + CC::TWL_TOTAL) * sizeof(V);
    static_assert(CR::DATA_ELEMS == CC::DATA_ELEMS && PLANE_LDS <= 160 * 1024, "plane + twiddle tables must fit LDS");
#ifdef MIFFT_STATIC_LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[PLANE_LDS];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
    V* lds = (V*)smem;
    const int tid0 = threadIdx.x;
    V twr[1];
    fill_lds_tw<CR, 1>(lds + CR::DATA_ELEMS, (const V*)p.tw, tid0, p.inverse);
    if constexpr (!SHARED_TW) fill_lds_tw<CC, 1>(lds + CR::DATA_ELEMS + CSHIFT, (const V*)p.tlo, tid0, p.inverse);
    __syncthreads();

    constexpr long long PLANE = (long long)CR::N * CR::TILE;
    V pre[CR::PREFETCH ? CR::IPT(0) : 1][CR::R(0)];
    long long t = blockIdx.x;
    if constexpr (CR::PREFETCH) {
        if (t < p.n_tiles) load_pass0<CR>(p, pre, tile_id(p, t) * PLANE, CR::TILE, tid0);
    }
    const int tid_entry = tid0;
    for (; t < p.n_tiles; t += gridDim.x) {
        int tid = tid_entry;  // opaque per plane: see tile_kernel
#ifndef MIFFT_NO_OPAQUE_TID
        asm volatile("" : "+v"(tid));
#endif
        const long long base = tile_id(p, t) * PLANE;
        V cur[CR::PREFETCH ? CR::IPT(0) : 1][CR::R(0)];
        if constexpr (CR::PREFETCH) {
#pragma unroll
            for (int k = 0; k < CR::IPT(0); ++k)
#pragma unroll
                for (int j = 0; j < CR::R(0); ++j) cur[k][j] = pre[k][j];
            const long long tn = t + gridDim.x;
            if (tn < p.n_tiles) load_pass0<CR>(p, pre, tile_id(p, tn) * PLANE, CR::TILE, tid);
        }
        auto drain = [&]() {  // before the column side's HBM stores: see tile_kernel / run_pass
            if constexpr (CR::PREFETCH) vm_drain();
        };
        run_pass<CR, 0>(p, lds, twr, cur, base, CR::TILE, tid);  // rows: HBM -> ... -> LDS (natural order)
        V none[1][CC::R(0)];
        run_pass<CC, 0, CSHIFT>(p, lds, twr, none, base, CC::TILE, tid, 0, 0, drain);  // columns: LDS -> ... -> HBM
    }
}

// ---------------------------------------------------------------------------------------------
// plane_kernel_wp<CR, CC, PAD>: the fused plane with WAVE-PRIVATE exchanges.  plane_kernel above flattens the work items
// of every pass over the whole workgroup, so each of its six LDS exchanges needs two workgroup barriers and all 16 waves
// move through load / butterfly / LDS phases in lockstep (measured on 1280 planes of 128 x 128: VALU 24 % busy, LDS 22 %,
// HBM 55 %; removing the barriers -- wrong results, timing only -- saves 11-15 %).  Here a wave OWNS whole transforms:
//   row phase    wave w transforms rows [w*RPW, (w+1)*RPW) through all row passes; the exchanges between its passes touch
//                only its own rows of the LDS plane, and LDS instructions of one wave execute in order, so no barrier
//   hand-over    ONE workgroup barrier (rows complete -> columns may start)
//   column phase wave w transforms columns [w*CPW, (w+1)*CPW) through all column passes, again without a barrier; the last
//                pass stores to HBM (runs of CPW elements per row)
//   a second barrier after the last LDS read of the plane protects the plane buffer against the next plane's first write.
// Two barriers per plane instead of twelve, and the waves drift apart, so one wave's butterflies overlap another's LDS
// and HBM traffic.  LDS pitch N2 + PAD (PAD = 8 elements: four consecutive rows of a CPW-column block fall into four
// different 64-byte bank groups).  Row-phase exchanges keep the XOR swizzle of the rows configuration.
// Requirements: the configurations of plane_kernel; N1 % WAVES == 0, N2 % WAVES == 0, CPW a power of two <= 64, every pass
// an exact number of wave rounds.
// ---------------------------------------------------------------------------------------------
template <class CR, class CC, int PAD_>
struct WavePlane {
    static constexpr int N2 = CR::N, N1 = CC::N, THREADS = CR::THREADS, WAVES = THREADS / 64;
    static constexpr int RPW = N1 / WAVES, CPW = N2 / WAVES, BPL = 64 / (CPW > 0 ? CPW : 1);
    static constexpr int PITCH = N2 + PAD_;
    static constexpr int DATA = N1 * PITCH;
    static constexpr bool SHARED_TW = CR::N == CC::N && CR::NP == CC::NP && CR::R(0) == CC::R(0) && CR::R(1) == CC::R(1) &&
                                      CR::R(2) == CC::R(2) && CR::R(3) == CC::R(3);
    static constexpr int CSHIFT = SHARED_TW ? 0 : CR::TWL_TOTAL;
    static constexpr size_t LDS_BYTES = (size_t)(DATA + CSHIFT + CC::TWL_TOTAL) * 2 * sizeof(typename CR::T);
    static constexpr int RIPT(int i) { return RPW * CR::NB(i) / 64; }
    static constexpr int CIPT(int i) { return CC::NB(i) / BPL; }
    static constexpr bool exact() {
        if (THREADS % 64 || N1 % WAVES || N2 % WAVES || CPW < 1 || CPW > 64 || (CPW & (CPW - 1))) return false;
        for (int i = 0; i < CR::NP; ++i)
            if ((RPW * CR::NB(i)) % 64) return false;
        for (int i = 0; i < CC::NP; ++i)
            if (CC::NB(i) % BPL) return false;
        return true;
    }
};

// work item `sub` of a wave in row pass 0 -> (row of the wave rl, butterfly b).  FS (four-step rows, see plane_kernel_wp):
// the rows of the LDS plane are the COLUMNS of the [N2][N1] view of the transform, so adjacent lanes take adjacent rows
// (RPW x 8-byte runs in HBM) instead of adjacent butterflies
template <class CR, class CC, int PAD, bool FS>
MIFFT_DEV void wp_row_item0(int sub, int& rl, int& b) {
    using G = WavePlane<CR, CC, PAD>;
    if constexpr (FS) {
        rl = sub % G::RPW;
        b = sub / G::RPW;
    } else {
        rl = sub / CR::NB(0);
        b = sub - rl * CR::NB(0);
    }
}

template <class CR, class CC, int PAD, int PART = 0, int NPARTS = 1, bool FS = false>
MIFFT_DEV void wp_load_rows(const TileParams& p, cpx<typename CR::T> (*v)[CR::R(0)], long long base, int wave, int lane) {
    using G = WavePlane<CR, CC, PAD>;
    using V = cpx<typename CR::T>;
    using T = typename CR::T;
    constexpr int R = CR::R(0), NB = CR::NB(0), IPT = G::RIPT(0);
    const V* gin = (const V*)p.in;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int rl, b;
        wp_row_item0<CR, CC, PAD, FS>(k * 64 + lane, rl, b);
        if constexpr (FS) {  // element (row rho, column gamma) of the plane = x[gamma * N1 + rho]
            static_assert(!FS || same_t<typename CR::IT, T>::value, "four-step rows: input of the plan's dtype");
            const unsigned off = (unsigned)b * (unsigned)G::N1 + (unsigned)(wave * G::RPW + rl);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if ((k * R + j) % NPARTS != PART) continue;
                if constexpr (CR::IN_REAL) {
                    v[k][j].x = gload_real<(CR::NT & 1) != 0>((const T*)p.in + base + (long long)j * NB * G::N1 + off);
                    v[k][j].y = (T)0;
                } else {
                    v[k][j] = gload<(CR::NT & 1) != 0>(gin + base + (long long)j * NB * G::N1 + off);
                }
            }
            continue;
        }
        const unsigned off = (unsigned)(wave * G::RPW + rl) * (unsigned)G::N2 + (unsigned)b;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if ((k * R + j) % NPARTS != PART) continue;  // (compile-time after unrolling)
            if constexpr (!same_t<typename CR::IT, T>::value) {
                v[k][j] = load_foreign<CR>(p.in, base + j * NB + off);
            } else if constexpr (CR::IN_REAL) {  // real tensor promoted in the load (load_pass0)
                v[k][j].x = gload_real<(CR::NT & 1) != 0>((const T*)p.in + base + j * NB + off);
                v[k][j].y = (T)0;
            } else {
                v[k][j] = gload<(CR::NT & 1) != 0>(gin + base + j * NB + off);
            }
        }
    }
}

template <class CR, class CC, int PAD, int I, class Hook, bool FS = false>
MIFFT_DEV void wp_row_passes(const TileParams& p, cpx<typename CR::T>* lds, cpx<typename CR::T> (*pre)[CR::R(0)], int wave,
                             int lane, Hook between) {
    if constexpr (I < CR::NP) {
        using G = WavePlane<CR, CC, PAD>;
        using T = typename CR::T;
        using V = cpx<T>;
        constexpr int R = CR::R(I), P = CR::P(I), NB = CR::NB(I), IPT = G::RIPT(I);
        const V* ltw = lds + G::DATA;
        V v[IPT][R];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            int rl, b;
            if constexpr (I == 0) {
                wp_row_item0<CR, CC, PAD, FS>(k * 64 + lane, rl, b);  // as in wp_load_rows
            } else {
                const int sub = k * 64 + lane;
                rl = sub / NB;
                b = sub - rl * NB;
            }
            V* row = lds + (wave * G::RPW + rl) * G::PITCH;
            if constexpr (I == 0) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    v[k][j] = pre[k][j];
                    if (p.inverse && !CR::IN_REAL) v[k][j].y = -v[k][j].y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) v[k][j] = row[swz<CR, I - 1>(b + j * NB)];
                const int pp = b % P;
#pragma unroll
                for (int j = 1; j < R; ++j) v[k][j] = cmul(v[k][j], ltw[CR::TWL_OFF(I) + (j - 1) * P + pp]);
            }
        }
        if constexpr (I > 0) wave_lds_fence();  // this wave's reads of the exchange precede its writes below
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            int rl, b;
            if constexpr (I == 0) {
                wp_row_item0<CR, CC, PAD, FS>(k * 64 + lane, rl, b);
            } else {
                const int sub = k * 64 + lane;
                rl = sub / NB;
                b = sub - rl * NB;
            }
            V* row = lds + (wave * G::RPW + rl) * G::PITCH;
            Dft<R, T, 1>::run(v[k]);
            const int q = b / P, pp = b - q * P, o0 = q * P * R + pp;
#pragma unroll
            for (int s = 0; s < R; ++s) {
                if constexpr (I == CR::NP - 1) {
                    if constexpr (FS) {
                        // four-step twiddle between the two sides: W_M^(rho * kappa), M = N1 * N2, from the two-level
                        // table behind the plane (lo: W_M^l, l < N2; hi: W_M^(N2 h) = W_N1^h, h < N1)
                        const V* fs = lds + G::LDS_BYTES / sizeof(V);
                        const int m = (wave * G::RPW + rl) * (o0 + s * P);
                        v[k][s] = cmul(v[k][s], cmul(fs[m % G::N2], fs[G::N2 + m / G::N2]));
                    }
                    row[o0 + s * P] = v[k][s];  // hand-over layout: natural order
                } else {
                    row[swz<CR, I>(o0 + s * P)] = v[k][s];
                }
            }
        }
        wave_lds_fence();
        if constexpr (I == 0) between(IntC<1>{});
        wp_row_passes<CR, CC, PAD, I + 1, Hook, FS>(p, lds, pre, wave, lane, between);
    }
}

// WL ("wide last"): the LAST column pass is not wave-private.  Behind one more workgroup barrier its butterflies are dealt
// over the whole workgroup with lanes along the contiguous axis, so that a wave's store instruction writes 64 adjacent
// columns of one row (512-byte runs) instead of CPW columns of 64 / CPW rows (64-byte runs at CPW = 8: every 128-byte line
// written half by one wave and half by its neighbour).  Three workgroup barriers per plane instead of two.
template <class CR, class CC, int PAD, int I, class Hook, class Hook2, bool WL = false>
MIFFT_DEV void wp_col_passes(const TileParams& p, cpx<typename CR::T>* lds, long long base, int wave, int lane,
                             Hook before_stores, Hook2 between) {
    if constexpr (WL && I == CC::NP - 1 && I > 0) {
        using G = WavePlane<CR, CC, PAD>;
        using T = typename CR::T;
        using V = cpx<T>;
        constexpr int R = CC::R(I), P = CC::P(I), NB = CC::NB(I);
        static_assert((NB * G::N2) % G::THREADS == 0 && G::THREADS % G::N2 == 0, "whole sweeps of the last column pass");
        constexpr int KPT = NB * G::N2 / G::THREADS, BSTEP = G::THREADS / G::N2;
        const V* ltw = lds + G::DATA + G::CSHIFT;
        const int tid = wave * 64 + lane;
        const int c = tid % G::N2, b0 = tid / G::N2;
        __syncthreads();  // every wave's scatter of pass NP-2 is complete
        V v[KPT][R];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int b = b0 + k * BSTEP;
#pragma unroll
            for (int j = 0; j < R; ++j) v[k][j] = lds[(b + j * NB) * G::PITCH + c];
            const int pp = b % P;
#pragma unroll
            for (int j = 1; j < R; ++j) v[k][j] = cmul(v[k][j], ltw[CC::TWL_OFF(I) + (j - 1) * P + pp]);
        }
        MIFFT_STAMP(G, 4);
        __syncthreads();  // last LDS read of this plane
        MIFFT_STAMP(G, 5);
        before_stores();
        MIFFT_STAMP(G, 6);
        V* gout = (V*)p.out;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int b = b0 + k * BSTEP;
            Dft<R, T, 1>::run(v[k]);
            const int q = b / P, pp = b - q * P, o0 = q * P * R + pp;
            const unsigned off = (unsigned)o0 * (unsigned)G::N2 + (unsigned)c;
#pragma unroll
            for (int s = 0; s < R; ++s) {
                if (CC::HS && s * P > CC::N / 2) continue;  // (never stored: not computed either, see pass_compute_scatter)
                V y = v[k][s];
                if (p.inverse) {
                    y.x *= (T)p.scale;
                    y.y *= -(T)p.scale;
                }
                if (!CC::HS || o0 + s * P <= p.store_lim) gstore<(CC::NT & 2) != 0>(gout + base + (long long)s * P * G::N2 + off, y);
            }
        }
    } else if constexpr (I < CC::NP) {
        using G = WavePlane<CR, CC, PAD>;
        using T = typename CR::T;
        using V = cpx<T>;
        constexpr int R = CC::R(I), P = CC::P(I), NB = CC::NB(I), IPT = G::CIPT(I);
        const V* ltw = lds + G::DATA + G::CSHIFT;
        const int cl = lane % G::CPW, bl = lane / G::CPW;
        V* col = lds + wave * G::CPW + cl;
        V v[IPT][R];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int b = bl + k * G::BPL;
#pragma unroll
            for (int j = 0; j < R; ++j) v[k][j] = col[(b + j * NB) * G::PITCH];
            if constexpr (I > 0) {
                const int pp = b % P;
#pragma unroll
                for (int j = 1; j < R; ++j) v[k][j] = cmul(v[k][j], ltw[CC::TWL_OFF(I) + (j - 1) * P + pp]);
            }
        }
        if constexpr (I == CC::NP - 1) {
            MIFFT_STAMP(G, 4);  // column passes up to the last gather
            __syncthreads();  // last LDS read of this plane: the next plane's row passes may overwrite the buffer
            MIFFT_STAMP(G, 5);  // barrier behind the last gather
            before_stores();  // wait for the prefetched plane ahead of the HBM stores (run_pass)
            MIFFT_STAMP(G, 6);  // drain: wait for the prefetched plane
        } else {
            wave_lds_fence();
        }
        V* gout = (V*)p.out;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int b = bl + k * G::BPL;
            Dft<R, T, 1>::run(v[k]);
            const int q = b / P, pp = b - q * P, o0 = q * P * R + pp;
            if constexpr (I == CC::NP - 1) {
                const unsigned off = (unsigned)o0 * (unsigned)G::N2 + (unsigned)(wave * G::CPW + cl);
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    if (CC::HS && s * P > CC::N / 2) continue;
                    V y = v[k][s];
                    if (p.inverse) {
                        y.x *= (T)p.scale;
                        y.y *= -(T)p.scale;
                    }
                    if (!CC::HS || o0 + s * P <= p.store_lim)
                        gstore<(CC::NT & 2) != 0>(gout + base + (long long)s * P * G::N2 + off, y);
                }
            } else {
#pragma unroll
                for (int s = 0; s < R; ++s) col[(o0 + s * P) * G::PITCH] = v[k][s];
            }
        }
        if constexpr (I < CC::NP - 1) wave_lds_fence();
        if constexpr (I == 0 && CC::NP > 1) between(IntC<3>{});
        wp_col_passes<CR, CC, PAD, I + 1, Hook, Hook2, WL>(p, lds, base, wave, lane, before_stores, between);
    }
}

// FS = true: the same kernel as a ONE-DIMENSIONAL transform of M = N1 * N2 points per "plane" (four-step inside LDS):
//   X[N2 kr + kc] = sum_rho W_N1^(rho kr) * W_M^(rho kc) * [ sum_gamma x[N1 gamma + rho] W_N2^(gamma kc) ]
// i.e. row rho of the plane holds x[N1 gamma + rho] (the loads transpose: adjacent lanes take adjacent rows), the row side
// transforms over gamma, its last pass multiplies by W_M^(rho kc), the column side transforms over rho and stores the
// result in natural order.  Twiddles of one 128-point side + a two-level table of N1 + N2 entries replace the M-entry
// table a one-row-per-workgroup tile kernel reads from L2 in every pass (p.thi = the M-entry table W_M^m, forward).
template <class CR, class CC, int PAD, bool FS = false, bool WL = false>
__global__ __launch_bounds__(CR::THREADS, CR::MINW) void plane_kernel_wp(const TileParams p) {
    using G = WavePlane<CR, CC, PAD>;
    using T = typename CR::T;
    using V = cpx<T>;
    static_assert(!CR::COLS && CC::COLS && CR::N == CC::TILE && CR::TILE == CC::N, "plane geometry");
    static_assert(CR::THREADS == CC::THREADS && CR::TWMODE == TW_LDS && CC::TWMODE == TW_LDS, "one thread count, LDS twiddles");
    static_assert(G::exact(), "every pass must be an exact number of wave rounds over wave-owned rows / columns");
    static_assert(G::LDS_BYTES <= 160 * 1024, "plane + twiddle tables must fit LDS");
    constexpr size_t FS_BYTES = FS ? (size_t)(G::N1 + G::N2) * sizeof(V) : 0;  // two-level four-step table behind the plane
    static_assert(G::LDS_BYTES + FS_BYTES <= 160 * 1024 && G::LDS_BYTES % sizeof(V) == 0, "four-step table must fit too");
#ifdef MIFFT_STATIC_LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES + FS_BYTES];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
    V* lds = (V*)smem;
    const int tid0 = threadIdx.x;
    fill_lds_tw<CR, 1>(lds + G::DATA, (const V*)p.tw, tid0, p.inverse);
    if constexpr (!G::SHARED_TW) fill_lds_tw<CC, 1>(lds + G::DATA + G::CSHIFT, (const V*)p.tlo, tid0, p.inverse);
    if constexpr (FS) {
        V* fs = lds + G::LDS_BYTES / sizeof(V);
        const V* wm = (const V*)p.thi;
        for (int e = tid0; e < G::N2 + G::N1; e += CR::THREADS) fs[e] = e < G::N2 ? wm[e] : wm[(e - G::N2) * G::N2];
    }
    __syncthreads();

    constexpr long long PLANE = (long long)G::N1 * G::N2;
    MIFFT_STAMP(G, -1);
    V pre[G::RIPT(0)][CR::R(0)];
    long long t = blockIdx.x;
    if (t < p.n_tiles) wp_load_rows<CR, CC, PAD, 0, 1, FS>(p, pre, tile_id(p, t) * PLANE, tid0 >> 6, tid0 & 63);
    for (; t < p.n_tiles; t += gridDim.x) {
        int tid = tid0;  // opaque per plane: offsets are re-derived instead of being hoisted and kept live (tile_kernel)
#ifndef MIFFT_NO_OPAQUE_TID
        asm volatile("" : "+v"(tid));
#endif
        const int wave = tid >> 6, lane = tid & 63;
        const long long base = tile_id(p, t) * PLANE;
        V cur[G::RIPT(0)][CR::R(0)];
#pragma unroll
        for (int k = 0; k < G::RIPT(0); ++k)
#pragma unroll
            for (int j = 0; j < CR::R(0); ++j) cur[k][j] = pre[k][j];
        const long long tn = t + gridDim.x;
        // the next plane's loads in four slices: here, behind the first row exchange, behind the hand-over barrier and
        // behind the first column exchange (see run_pass: a burst of loads stalls every wave in its issue)
        constexpr int SLICES = (CR::PREFETCH && MIFFT_SLICED_PREFETCH_PLANE) ? 4 : 1;
        const long long nbase = tn < p.n_tiles ? tile_id(p, tn) * PLANE : 0;
        if (CR::PREFETCH && tn < p.n_tiles) wp_load_rows<CR, CC, PAD, 0, SLICES, FS>(p, pre, nbase, wave, lane);
        auto slice = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (CR::PREFETCH && SLICES > 1 && K >= 1 && K < SLICES) {
                if (tn < p.n_tiles) wp_load_rows<CR, CC, PAD, K, SLICES, FS>(p, pre, nbase, wave, lane);
            }
        };
        auto drain = [&]() {  // ahead of this plane's HBM stores (tile_kernel / run_pass)
            if constexpr (CR::PREFETCH) vm_drain();
        };
        MIFFT_STAMP(G, 0);  // plane bookkeeping, take-over of the prefetched registers, first slice of the next loads
        wp_row_passes<CR, CC, PAD, 0, decltype(slice), FS>(p, lds, cur, wave, lane, slice);
        MIFFT_STAMP(G, 1);  // row passes (wave-private)
        __syncthreads();  // hand-over: every row is complete before any column starts
        MIFFT_STAMP(G, 2);  // hand-over barrier
        slice(IntC<2>{});
        wp_col_passes<CR, CC, PAD, 0, decltype(drain), decltype(slice), WL>(p, lds, base, wave, lane, drain, slice);
        MIFFT_STAMP(G, 3);  // column passes incl. the barrier behind the last gather and the HBM stores
        if (!CR::PREFETCH && tn < p.n_tiles) wp_load_rows<CR, CC, PAD, 0, 1, FS>(p, pre, tile_id(p, tn) * PLANE, wave, lane);
    }
    MIFFT_STAMP_DUMP(G, p);
}

#ifdef MIFFT_EXPERIMENTAL  // lab builds only: a documented negative result (DESIGN_EXPERIMENTS.md), not in libmifft.so
// ---------------------------------------------------------------------------------------------
// image_kernel<CR, CC>: the two innermost dimensions (N1 x N2, N2 contiguous) of images that do NOT fit LDS but fit
// one XCD's 4-MB L2 (100 x 640 x 480: 2.4 MB each).  Every XCD owns whole images: its workgroups transform the rows
// of an image (x -> out, configuration CR), meet at an XCD-LOCAL barrier, and transform the columns in place
// (configuration CC) while the row results are still in that XCD's L2 -- the column pass reads L2 instead of HBM, and
// row results that are overwritten before they are evicted never reach HBM at all (guide: "each XCD has its own L2,
// so make the blockIdx -> tile mapping XCD-aware").
//   * XCD of a workgroup: the hardware XCC_ID register, not an assumption about the dispatch order; the slot inside the
//     XCD is a ticket from an atomic counter that lives in that XCD's L2.
//   * barrier: one arrival counter per XCD, relaxed agent-scope atomics (executed in the L2 all participants share).
//     The stores of a phase are complete in L2 before the arrival is counted (__syncthreads() drains vmcnt); an
//     agent-scope RELEASE fence is deliberately not used -- on gfx942/950 it writes the L2 back to HBM.
//   * every spin is bounded (a dispatch that does not give each XCD its workgroups ends with a wrong result instead of
//     a hang); the host probes the XCC_ID distribution of the same launch geometry at plan time before it selects this
//     kernel (kernels_jit.cpp).
// Launch: 8 * wgs_per_xcd workgroups, all resident (one per CU); counters zeroed before every launch.
// ---------------------------------------------------------------------------------------------
struct ImageParams {
    const void* in;
    void* out;
    const void* tw_rows;  // W_N2
    const void* tw_cols;  // W_N1
    long long n_images;
    unsigned* counters;  // [0..7] tickets, [8..15] arrivals (zeroed before every launch), [16] sticky error flags
    int wgs_per_xcd;
    int inverse;
};

MIFFT_DEV unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// error flags of the image kernel (ImageParams::counters[16]; read and cleared by mifft_plan_device_status)
enum { IMAGE_ERR_SPIN_EXPIRED = 1u, IMAGE_ERR_SURPLUS_WORKGROUP = 2u };

MIFFT_DEV void xcd_barrier(unsigned* arrive, unsigned target, int tid, unsigned* err) {
    __syncthreads();  // workgroup-scope release: every wave's global stores of the phase are complete in L2
    if (tid == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 20))
            __builtin_amdgcn_s_sleep(8);
        // the bounded spin gave up: this workgroup goes on WITHOUT its XCD's row results being complete, so the output
        // of this exec is not to be trusted -- say so where the host can see it
        if (spins >= (1u << 20)) __hip_atomic_fetch_or(err, (unsigned)IMAGE_ERR_SPIN_EXPIRED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's L1 lines; nothing is written back
}

template <class CR, class CC>
__global__ __launch_bounds__(CR::THREADS, 1) void image_kernel(const ImageParams q) {
    using T = typename CR::T;
    using V = cpx<T>;
    static_assert(!CR::COLS && CC::COLS && CR::THREADS == CC::THREADS, "rows configuration, columns configuration");
    static_assert(CR::FIRST_DIRECT && CR::LAST_DIRECT && CC::FIRST_DIRECT && CC::LAST_DIRECT, "direct tiles");
    static_assert(CR::TWMODE == TW_LDS && CC::TWMODE == TW_LDS && !CR::PREFETCH && !CC::PREFETCH, "LDS twiddles");
    static_assert(!CR::BIGP0 && !CC::BIGP0 && !CC::TSTORE, "register butterflies only");
    constexpr int MAXD = CR::DATA_ELEMS > CC::DATA_ELEMS ? CR::DATA_ELEMS : CC::DATA_ELEMS;
    constexpr int TWS_R = MAXD - CR::DATA_ELEMS, TWS_C = MAXD + CR::TWL_TOTAL - CC::DATA_ELEMS;
    constexpr size_t IMAGE_LDS = (size_t)(MAXD + CR::TWL_TOTAL + CC::TWL_TOTAL) * sizeof(V);
    static_assert(IMAGE_LDS <= 160 * 1024, "tiles + both twiddle tables must fit LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[IMAGE_LDS];
    __shared__ unsigned s_slot;
    V* lds = (V*)smem;
    const int tid0 = threadIdx.x;
    constexpr long long N1 = CC::N, N2 = CR::N;

    fill_lds_tw<CR, 1>(lds + MAXD, (const V*)q.tw_rows, tid0, q.inverse);
    fill_lds_tw<CC, 1>(lds + MAXD + CR::TWL_TOTAL, (const V*)q.tw_cols, tid0, q.inverse);
    const unsigned xcc = xcc_id() & 7;
    if (tid0 == 0) s_slot = __hip_atomic_fetch_add(q.counters + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long long slot = s_slot, W = q.wgs_per_xcd;
    unsigned* arrive = q.counters + 8 + xcc;
    if (slot >= W) {  // more workgroups on this XCD than planned (the plan-time probe rules this out): another XCD is
                      // short of workgroups and its barrier cannot complete -- flag the exec and take no part
        if (tid0 == 0) __hip_atomic_fetch_or(q.counters + 16, (unsigned)IMAGE_ERR_SURPLUS_WORKGROUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    TileParams pr{}, pc{};
    pr.in = q.in;
    pr.out = q.out;
    pr.tw = q.tw_rows;
    pr.inner = 1;
    pr.inverse = q.inverse;
    pr.scale = q.inverse ? 1.0 / (double)N2 : 1.0;
    pc.in = q.out;
    pc.out = q.out;
    pc.tw = q.tw_cols;
    pc.inner = N2;
    pc.inverse = q.inverse;
    pc.scale = q.inverse ? 1.0 / (double)N1 : 1.0;
    constexpr long long TILES_R = (N1 + CR::TILE - 1) / CR::TILE, TILES_C = (N2 + CC::TILE - 1) / CC::TILE;
    V twr[1];
    V none_r[1][CR::R(0)], none_c[1][CC::R(0)];
    unsigned phase = 0;
    for (long long img = xcc; img < q.n_images; img += 8) {
        const long long ibase = img * N1 * N2;
        for (long long tr = slot; tr < TILES_R; tr += W) {  // rows of this image: x -> out
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const long long left = N1 - tr * CR::TILE;
            run_pass<CR, 0, TWS_R>(pr, lds, twr, none_r, ibase + tr * CR::TILE * N2, (int)(left < CR::TILE ? left : CR::TILE), tid);
            __syncthreads();
        }
        xcd_barrier(arrive, (unsigned)((++phase) * W), tid0, q.counters + 16);
        for (long long tc = slot; tc < TILES_C; tc += W) {  // columns, in place, from this XCD's L2
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const long long left = N2 - tc * CC::TILE;
            run_pass<CC, 0, TWS_C>(pc, lds, twr, none_c, ibase + tc * CC::TILE, (int)(left < CC::TILE ? left : CC::TILE), tid);
            __syncthreads();
        }
        // no barrier here: the next image is other memory, and every workgroup still arrives exactly once per barrier
    }
}

#endif  // MIFFT_EXPERIMENTAL

// persistent grid: enough workgroups to fill every CU to its LDS / wave limit.  The formula ignores registers, so a kernel
// that holds fewer workgroups than it launches leaves the surplus waiting for a slot; whether that hurts is kernel by kernel
// (rows480: 4 launched / 3 resident 0.0873 ms, 3 launched 0.0834 ms; rows128: 8 launched / 5 resident is 2-7 % FASTER than
// 5 launched -- the late workgroups fill the holes of the ragged end), so measured per-kernel values override the formula
// (`wg_per_cu_override`, from kGridPerCu in kernels_fast.hip) instead of a general occupancy rule.
template <class C>
inline long long tile_grid(int num_cus, long long n_tiles, int wg_per_cu_override = 0) {
    long long per_cu = (160 * 1024) / (long long)(C::LDS_BYTES ? C::LDS_BYTES : 1);
    const long long wave_limit = 2048 / C::THREADS;
    if (per_cu > wave_limit) per_cu = wave_limit;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    if (wg_per_cu_override > 0) per_cu = wg_per_cu_override;
    long long grid = (long long)num_cus * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    return grid < 1 ? 1 : grid;
}

}  // namespace mifft
