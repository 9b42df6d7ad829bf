// tile_kernel_experimental.h -- kernels that were measured on MI355X and NOT shipped ("negative
// results"): the LDS-DMA staged variant of the flat-copy tile kernel and its output-split radix-31 pass 0.  They are
// kept here, outside the product sources, so that the tuner (tune_tile.hip) can re-run the experiments.
#pragma once
#include "tile_kernel.h"

namespace mifft {

// ---------------------------------------------------------------------------------------------
// tile_kernel_dma<C>: the flat-copy configurations (rows that are not 16-byte aligned, e.g. N = 93)
// with an asynchronous HBM -> LDS copy.  While the passes of tile t run in the work buffer, the
// rows of tile t+1 stream into a staging buffer by LDS-DMA (global_load_lds_dwordx4: per-lane
// source address, LDS destination = wave-uniform base + lane*16, no VGPRs), so the workgroup
// always has a whole tile of HBM reads in flight.  Pass 0 reads the staging buffer and scatters
// into the work buffer; the copy of the next tile is issued as soon as every wave has gathered.
// Requires a 16-byte aligned tile base (checked by the launcher; otherwise tile_kernel<C> runs).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* mifft_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* mifft_glb_ptr_t;

template <class C>
MIFFT_DEV void dma_issue_tile(const TileParams& p, cpx<typename C::T>* stage, long long base, int nv, int tid) {
    using V = cpx<typename C::T>;
    constexpr int PER16 = 16 / (int)sizeof(V);            // complex elements per 16-byte piece
    const int pieces = (nv * C::N) / PER16;               // launcher guarantees nv*N*sizeof(V) % 16 == 0
    const char* g = (const char*)((const V*)p.in + base);
    const int lane = tid & 63, wave0 = tid - lane;  // first thread of this wave (the last wave may be partial)
    for (int w0 = wave0; w0 < pieces; w0 += C::THREADS) {
        const int idx = w0 + lane;
        if (idx < pieces)
            __builtin_amdgcn_global_load_lds((mifft_glb_ptr_t)(g + (size_t)idx * 16),
                                             (mifft_lds_ptr_t)((char*)stage + (size_t)w0 * 16), 16, 0, 0);
    }
    // odd element count (a ragged last tile of odd-length rows): the trailing 8 bytes go the ordinary way
    if (PER16 == 2 && ((nv * C::N) & 1) && tid == 0) stage[nv * C::N - 1] = ((const V*)p.in)[base + nv * C::N - 1];
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void tile_kernel_dma(const TileParams p) {
    using T = typename C::T;
    using V = cpx<T>;
    static_assert(C::DMA && !C::COLS && !C::FIRST_DIRECT && !C::LAST_DIRECT && !C::IN_REAL && C::LD == C::N,
                  "DMA staging is for the flat-copy row configurations");
    static_assert(C::TWMODE != TW_REG, "register twiddles not wired for the DMA variant");
    // Two DISTINCT LDS objects: the module-LDS lowering gives each its own alias scope, which lets the
    // waitcnt pass see that LDS writes into the work buffer do not touch the DMA destination -- with one
    // (dynamic) LDS array it put s_waitcnt vmcnt(0) in front of the first ds_write after the DMA issue.
    __shared__ __attribute__((aligned(16))) V s_work[C::DATA_ELEMS + (C::TWL_TOTAL > 0 ? C::TWL_TOTAL : 1)];
    __shared__ __attribute__((aligned(16))) V s_stage[C::STAGE_ELEMS];
    V* lds = s_work;
    V* stage = s_stage;
    const int tid = threadIdx.x;
    V twr[1];
    if constexpr (C::TWMODE == TW_LDS) fill_lds_tw<C, 1>(lds + C::DATA_ELEMS, (const V*)p.tw, tid, p.inverse);

    long long t = blockIdx.x;
    if (t < p.n_tiles) {
        long long base;
        int nv;
        tile_geom<C>(p, t, base, nv);
        dma_issue_tile<C>(p, stage, base, nv, tid);
    }
    for (; t < p.n_tiles; t += gridDim.x) {
        long long base;
        int nv;
        tile_geom<C>(p, t, base, nv);
        // this wave's share of the copy has landed (and its previous stores have drained) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier<C>();  // ... and so has every other wave's; nobody still reads the work buffer
        constexpr int R0 = C::R(0), IPT0 = C::IPT(0);
        V v[IPT0][R0];
        pass_gather_lds<C, 0>(p, stage, lds + C::DATA_ELEMS, twr, v, tid);
        if (p.inverse) {
#pragma unroll
            for (int k = 0; k < IPT0; ++k)
#pragma unroll
                for (int j = 0; j < R0; ++j) v[k][j].y = -v[k][j].y;
        }
        wg_barrier<C>();  // staging buffer fully consumed
        const long long tn = t + gridDim.x;
        if (tn < p.n_tiles) {
            long long nbase;
            int nnv;
            tile_geom<C>(p, tn, nbase, nnv);
            dma_issue_tile<C>(p, stage, nbase, nnv, tid);  // overlaps with everything below
        }
        pass_compute_scatter<C, 0>(p, lds, v, base, nv, tid);
        wg_barrier<C>();
        V none[1][R0];
        run_pass<C, 1>(p, lds, twr, none, base, nv, tid);
        // flat coalesced LDS -> HBM store of the finished tile
        V* gout = (V*)p.out;
        const int total = nv * C::N;
        for (int f = tid; f < total; f += C::THREADS) {
            V y = lds[f];  // last exchange is natural order, LD == N
            if (p.inverse) {
                y.x *= (T)p.scale;
                y.y *= -(T)p.scale;
            }
            gstore<(C::NT & 2) != 0>(gout + base + f, y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile_kernel_dma_split<C, G>: tile_kernel_dma with an OUTPUT-SPLIT prime pass 0.  A row of 93 = 31 * 3
// points has only three 31-point butterflies, each ~170 live VGPRs: too few, too fat work items to hide
// their latency (with a free butterfly the DMA kernel reaches 0.131 ms at 500k x 93, with the real one
// 0.167 ms).  Here G lanes share one butterfly: lane group g reads the 31 staged inputs, forms the
// conjugate sums and computes only its own output pairs, writing each to the work buffer at once.
// Slots are laid out [g][padded item] with the pad a multiple of 64, so g is wave-uniform and its
// constants stay compile-time.
// ---------------------------------------------------------------------------------------------
template <class C, int G, int g0, class X, class Emit>
MIFFT_DEV void prime_group_dispatch(int g, const X* x, Emit& emit) {
    if constexpr (g0 < G) {
        if (g == g0)
            PrimeGroup<C::R(0), typename C::T, G, g0>::run(x, emit);
        else
            prime_group_dispatch<C, G, g0 + 1>(g, x, emit);
    }
}

template <class C, int G>
__global__ __launch_bounds__(C::THREADS, C::MINW) void tile_kernel_dma_split(const TileParams p) {
    using T = typename C::T;
    using V = cpx<T>;
    static_assert(C::DMA && !C::COLS && !C::FIRST_DIRECT && !C::LAST_DIRECT && !C::IN_REAL && C::LD == C::N,
                  "DMA staging is for the flat-copy row configurations");
    static_assert(C::TWMODE != TW_REG && is_prime_ce(C::R(0)) && C::R(0) > 2 && C::THREADS % 64 == 0, "split prime pass 0");
    __shared__ __attribute__((aligned(16))) V s_work[C::DATA_ELEMS + (C::TWL_TOTAL > 0 ? C::TWL_TOTAL : 1)];
    __shared__ __attribute__((aligned(16))) V s_stage[C::STAGE_ELEMS];
    V* lds = s_work;
    V* stage = s_stage;
    const int tid = threadIdx.x;
    V twr[1];
    if constexpr (C::TWMODE == TW_LDS) fill_lds_tw<C, 1>(lds + C::DATA_ELEMS, (const V*)p.tw, tid, p.inverse);

    constexpr int R0 = C::R(0), NB0 = C::NB(0);
    constexpr int PER_G = NB0 * C::TILE, PAD = (PER_G + 63) / 64 * 64, SLOTS = PAD * G;
    constexpr int ROUNDS = (SLOTS + C::THREADS - 1) / C::THREADS;

    long long t = blockIdx.x;
    if (t < p.n_tiles) {
        long long base;
        int nv;
        tile_geom<C>(p, t, base, nv);
        dma_issue_tile<C>(p, stage, base, nv, tid);
    }
    for (; t < p.n_tiles; t += gridDim.x) {
        long long base;
        int nv;
        tile_geom<C>(p, t, base, nv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier<C>();
        // ---- pass 0: staged rows -> G lanes per prime butterfly -> work buffer ----
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int id = tid + k * C::THREADS;
            if (SLOTS % C::THREADS == 0 || id < SLOTS) {
                const int g = __builtin_amdgcn_readfirstlane(id / PAD);  // PAD and THREADS are multiples of 64
                const int r = id - g * PAD;
                if (r < PER_G) {
                    const int c = r / NB0, b = r - c * NB0;
                    V x[R0];
#pragma unroll
                    for (int j = 0; j < R0; ++j) x[j] = stage[c * C::N + b + j * NB0];
                    if (p.inverse) {
#pragma unroll
                        for (int j = 0; j < R0; ++j) x[j].y = -x[j].y;
                    }
                    V* row = lds + c * C::LD;
                    auto emit = [&](int s, V val) { row[swz<C, 0>(b * R0 + s)] = val; };  // P = 1: position q*R + s
                    prime_group_dispatch<C, G, 0>(g, x, emit);
                }
            }
        }
        wg_barrier<C>();  // staging buffer consumed, work buffer holds pass-0 output
        const long long tn = t + gridDim.x;
        if (tn < p.n_tiles) {
            long long nbase;
            int nnv;
            tile_geom<C>(p, tn, nbase, nnv);
            dma_issue_tile<C>(p, stage, nbase, nnv, tid);
        }
        V none[1][R0];
        run_pass<C, 1>(p, lds, twr, none, base, nv, tid);
        V* gout = (V*)p.out;
        const int total = nv * C::N;
        for (int f = tid; f < total; f += C::THREADS) {
            V y = lds[f];
            if (p.inverse) {
                y.x *= (T)p.scale;
                y.y *= -(T)p.scale;
            }
            gstore<(C::NT & 2) != 0>(gout + base + f, y);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// tile_kernel_pp<C>: PING-PONG column tiles.  A tile of a long strided dimension (16 columns x 640 points = 80 KB)
// leaves room for ONE workgroup per CU, and a workgroup that does everything in lockstep adds up its load time, its
// butterfly / LDS time and its store time (section "3.1b" of DESIGN.md).  Here the workgroup is two HALVES of
// C::THREADS threads, each transforming its own tiles with the configuration C, shifted by half a tile period, and the
// one LDS tile buffer changes hands between them:
//
//      half A:  | LDS phase: passes of tile 0        | memory phase: stores 0, loads 2      | LDS phase: tile 2 | ...
//      half B:  | memory phase: loads 1              | LDS phase: passes of tile 1          | memory: stores 1, loads 3 | ...
//
// so one half's HBM traffic always runs beside the other half's butterflies and LDS exchanges.  A tile needs the buffer
// from the scatter of pass 0 to the gather of the last pass; the last pass's butterflies and its HBM stores happen after
// the buffer has been handed over, from registers.
// Synchronisation is the workgroup barrier only.  Every PHASE has the same number of barriers for both halves
// (1 + 2 NP - 3): the LDS half executes "acquire" and the barriers between its passes, the memory half executes
// "release" (the same physical barrier as the other's acquire) and as many filler barriers.  Both halves run the same
// number of phases (tiles beyond the end are processed as shadows: no HBM access, same barriers), so the barrier counts
// match by construction and no wave can be left waiting.
// Requirements: column tile, direct first and last pass, no prefetch / WSUB / TSTORE / cooperative primes.
// MEASURED (tools/tune GROUP 22, 100 x 640 x 480 columns): 0.111 ms against 0.103 ms for the plain prefetching tile --
// with only four waves in each phase (one per SIMD) the LDS instructions and the butterflies run far below their
// rates (ds_*_b64 needs ~4 waves per SIMD), which costs more than the overlap brings.  Correct, not shipped.
// ---------------------------------------------------------------------------------------------
template <class C, int I>
MIFFT_DEV void pp_lds_passes(const TileParams& p, cpx<typename C::T>* lds, cpx<typename C::T> (*cur)[C::R(0)],
                             cpx<typename C::T> (*vlast)[C::R(C::NP - 1)], long long base, int nv, int htid) {
    using V = cpx<typename C::T>;
    const V* ltw = lds + C::DATA_ELEMS;
    if constexpr (I == 0) {
        V v[C::IPT(0)][C::R(0)];
#pragma unroll
        for (int k = 0; k < C::IPT(0); ++k)
#pragma unroll
            for (int j = 0; j < C::R(0); ++j) {
                v[k][j] = cur[k][j];
                if (p.inverse) v[k][j].y = -v[k][j].y;
            }
        pass_compute_scatter<C, 0>(p, lds, v, base, nv, htid);
        wg_barrier<C>();
        pp_lds_passes<C, 1>(p, lds, cur, vlast, base, nv, htid);
    } else if constexpr (I < C::NP - 1) {
        V v[C::IPT(I)][C::R(I)];
        pass_gather_lds<C, I>(p, lds, ltw, nullptr, v, htid);
        wg_barrier<C>();
        pass_compute_scatter<C, I>(p, lds, v, base, nv, htid);
        wg_barrier<C>();
        pp_lds_passes<C, I + 1>(p, lds, cur, vlast, base, nv, htid);
    } else {
        pass_gather_lds<C, I>(p, lds, ltw, nullptr, vlast, htid);  // the buffer is released at the next barrier
    }
}

template <class C>
__global__ __launch_bounds__(2 * C::THREADS, 1) void tile_kernel_pp(const TileParams p) {
    using T = typename C::T;
    using V = cpx<T>;
    static_assert(C::COLS && C::FIRST_DIRECT && C::LAST_DIRECT && !C::PREFETCH && !C::WSUB && !C::TSTORE && !C::FS1 &&
                      !C::BIGP0 && C::TWMODE == TW_LDS && C::NP >= 2 && C::THREADS % 64 == 0,
                  "ping-pong: a plain direct column tile with LDS twiddles");
#ifdef MIFFT_STATIC_LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
    V* lds = (V*)smem;
    constexpr int FILL = 2 * C::NP - 3;  // barriers between the passes of one LDS phase
    const int half = threadIdx.x / C::THREADS;  // wave-uniform
    const int htid0 = threadIdx.x - half * C::THREADS;
    // twiddle table: filled by both halves together
    for (int i = threadIdx.x; i < C::TWL_TOTAL; i += 2 * C::THREADS) lds[C::DATA_ELEMS + i] = V{(T)0, (T)0};
    __syncthreads();
    if (half == 0) fill_lds_tw<C, 1>(lds + C::DATA_ELEMS, (const V*)p.tw, htid0, p.inverse);
    __syncthreads();

    // tiles of this workgroup: t_k = blockIdx.x + k * gridDim.x, k < cnt; half h takes k = h, h + 2, ...
    const long long cnt = p.n_tiles > (long long)blockIdx.x ? (p.n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long long phases = 2 * ((cnt + 1) / 2);  // same for both halves
    V cur[C::IPT(0)][C::R(0)];               // pass-0 inputs of the tile about to enter its LDS phase
    V vlast[C::IPT(C::NP - 1)][C::R(C::NP - 1)];  // gathered inputs of the last pass of the tile that just left it
    long long base_ld = 0, base_st = 0;
    int nv_ld = 1, nv_st = 0;
    auto geom = [&](long long k, long long& base, int& nv) {  // false: shadow tile (beyond the end)
        if (k >= cnt) {
            nv = 0;
            return false;
        }
        tile_geom<C>(p, tile_id(p, (long long)blockIdx.x + k * gridDim.x), base, nv);
        return true;
    };
    if (half == 0) {  // prologue: half A's first tile
        if (geom(0, base_ld, nv_ld)) load_pass0<C>(p, cur, base_ld, nv_ld, htid0);
    }
    for (long long ph = 0; ph < phases; ++ph) {
        int htid = htid0;
        asm volatile("" : "+v"(htid));  // opaque per phase (see tile_kernel)
        if ((ph & 1) == half) {
            // ---- LDS phase of my tile k = ph (its loads were issued during my last memory phase / the prologue) ----
            wg_barrier<C>();  // acquire: the other half has gathered its last pass
            base_st = base_ld;
            nv_st = nv_ld;
            pp_lds_passes<C, 0>(p, lds, cur, vlast, base_st, nv_st, htid);
        } else {
            // ---- memory phase: finish my previous tile from registers, fetch my next one ----
            wg_barrier<C>();  // release (pairs with the other half's acquire)
            if (ph >= 1 && nv_st > 0) pass_compute_scatter<C, C::NP - 1>(p, lds, vlast, base_st, nv_st, htid);
            if (geom(ph + 1, base_ld, nv_ld)) load_pass0<C>(p, cur, base_ld, nv_ld, htid);
#pragma unroll
            for (int f = 0; f < FILL; ++f) wg_barrier<C>();  // the other half's barriers between its passes
        }
    }
    // the half that ran the last LDS phase (B: phases is even) releases and finishes its last tile
    wg_barrier<C>();
    if (half == 1 && phases > 0 && nv_st > 0) pass_compute_scatter<C, C::NP - 1>(p, lds, vlast, base_st, nv_st, threadIdx.x - C::THREADS);
}

}  // namespace mifft
