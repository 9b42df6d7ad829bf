// pipe_probe.hip -- which software pipeline lets ONE workgroup per CU (a 128-KiB tile that must pass through LDS as a
// whole, like the fused 128 x 128 plane) move its bytes at the speed of a flat copy?  A model of plane_kernel_wp: per tile
// PH phases of (LDS round trip + VALU work), all loads of the next tile prefetched into registers, results stored from
// registers.  Variants: load slices, DEFERRED stores (the results stay in registers and go out in slices at the seams of
// the NEXT tile), workgroup barriers on / off, threads, workgroups per CU.  Design evidence only.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o pipe_probe pipe_probe.hip && ./pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PART, int NPARTS, int E, int THREADS>
__device__ __forceinline__ void load_slice(f2 (&v)[E], const f2* g, long long es = THREADS) {
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e % NPARTS == PART) v[e] = g[e * es];
}
template <int PART, int NPARTS, int E, int THREADS>
__device__ __forceinline__ void store_slice(const f2 (&v)[E], f2* g, long long es = THREADS) {
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e % NPARTS == PART) g[e * es] = v[e];
}

// MODE 0: burst stores right after the last phase (today's kernels)   MODE 1: deferred, sliced stores
// W > 0: column tiles of W adjacent columns x (THREADS * E / W) rows of a [outer][rows][inner] tensor (in == out: in place)
template <int THREADS, int E, int PH, int LSL, int MODE, int WORK, bool BAR, int MINW, int W = 0>
__global__ __launch_bounds__(THREADS, MINW) void pipe(const f2* in, f2* out, long long n_tiles, int inner = 0) {
    extern __shared__ f2 lds[];
    constexpr long long TILE = (long long)THREADS * E;
    const int tid = threadIdx.x;
    f2 pre[E], outv[E];
    long long t = blockIdx.x;
    long long t_prev = -1;
    // element e of thread tid in tile t: flat: t * TILE + tid + e * THREADS; columns: row (e * THREADS + tid) / W, column tid % W
    const long long es = W > 0 ? (long long)(THREADS / (W > 0 ? W : 1)) * inner : THREADS;
    auto tbase = [&](long long tt) -> long long {
        if constexpr (W > 0) {
            const long long tpo = inner / W, o = tt / tpo, c0 = (tt - o * tpo) * W;
            return o * (TILE / W) * inner + c0 + (long long)(tid / W) * inner + tid % W;
        } else {
            return tt * TILE + tid;
        }
    };
    if (t < n_tiles) load_slice<0, 1, E, THREADS>(pre, in + tbase(t), es);
    for (; t < n_tiles; t += gridDim.x) {
        f2 cur[E];
        if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): prefetched tile + the previous tile's last stores
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = pre[e];
        const long long tn = t + gridDim.x;
        const f2* gn = in + tbase(tn < n_tiles ? tn : t);
        f2* gp = out + tbase(t_prev >= 0 ? t_prev : t);
        auto seam = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if (MODE == 1 && K >= 1 && t_prev >= 0) store_slice<K - 1, PH, E, THREADS>(outv, gp, es);  // seams 1..PH
            if (K < LSL && tn < n_tiles) load_slice<K, LSL, E, THREADS>(pre, gn, es);                   // seams 0..LSL-1
        };
        seam(std::integral_constant<int, 0>{});
        auto phase = [&](auto pc) {
            constexpr int P = decltype(pc)::value;
            // LDS round trip (a transposing exchange stands in for the Stockham scatter / gather)
#pragma unroll
            for (int e = 0; e < E; ++e) lds[e * THREADS + tid] = cur[e];
            if (BAR) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
            for (int e = 0; e < E; ++e) cur[e] = lds[e * THREADS + (tid ^ (BAR ? 65 : 1))];
            if (BAR) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            // butterflies: WORK dependent-free FMAs per element
#pragma unroll
            for (int w = 0; w < WORK; ++w)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    cur[e].x = __builtin_fmaf(cur[e].x, 1.0000001f, 1e-9f * (float)(w + P));
                    cur[e].y = __builtin_fmaf(cur[e].y, 0.9999999f, -1e-9f * (float)(w + P));
                }
        };
        phase(std::integral_constant<int, 0>{});
        seam(std::integral_constant<int, 1>{});
        if constexpr (PH > 1) { phase(std::integral_constant<int, 1>{}); seam(std::integral_constant<int, 2>{}); }
        if constexpr (PH > 2) { phase(std::integral_constant<int, 2>{}); seam(std::integral_constant<int, 3>{}); }
        if constexpr (PH > 3) { phase(std::integral_constant<int, 3>{}); seam(std::integral_constant<int, 4>{}); }
        if constexpr (MODE == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // the prefetched tile has landed (vm_drain of the product kernels)
            __builtin_amdgcn_sched_barrier(0);
            store_slice<0, 1, E, THREADS>(cur, out + tbase(t), es);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) outv[e] = cur[e];
            t_prev = t;
        }
    }
    if (MODE == 1 && t_prev >= 0) store_slice<0, 1, E, THREADS>(outv, out + tbase(t_prev), es);
}

template <int THREADS, int E, int PH, int LSL, int MODE, int WORK, bool BAR, int MINW, int W = 0>
void run(const char* name, const f2* a, f2* b, long long n_tiles, int wg_per_cu, int lds_bytes, int inner = 0) {
    auto k = pipe<THREADS, E, PH, LSL, MODE, WORK, BAR, MINW, W>;
    if (lds_bytes < THREADS * E * 8) lds_bytes = THREADS * E * 8;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, THREADS, lds_bytes));
    long long grid = 256LL * wg_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<<<(unsigned)grid, THREADS, lds_bytes>>>(a, b, n_tiles, inner);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) k<<<(unsigned)grid, THREADS, lds_bytes>>>(a, b, n_tiles, inner);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
        if (ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * THREADS * E * 8.0;
    printf("%-44s thr %4d E %2d ph %d lsl %d mode %d work %2d bar %d  wg/cu %d (occ %d) lds %6d: %7.4f ms %7.1f GB/s\n", name,
           THREADS, E, PH, LSL, MODE, WORK, (int)BAR, wg_per_cu, occ, lds_bytes, best, 2.0 * bytes / best / 1e6);
}

int main() {
    const long long elems = 10LL * 128 * 128 * 128;  // 10 x 128^3 complex64 = 167.8 MB
    f2 *a, *b;
    CK(hipMalloc(&a, elems * 8));
    CK(hipMalloc(&b, elems * 8));
    CK(hipMemset(a, 0, elems * 8));
    CK(hipMemset(b, 0, elems * 8));
    const long long planes = elems / 16384;
    const int L = 136 * 1024;
    // upper bound: no LDS reservation, small workgroups, no work
    run<256, 16, 1, 1, 0, 0, false, 1>("flat-ish: 4-KB chunks, 8 wg/cu", a, b, elems / 4096, 8, 0);
    // today's plane kernel as a model: 1024 threads, 3 phases, 4 load slices, burst stores, barriers
    run<1024, 16, 3, 1, 0, 0, true, 4>("plane model: burst loads, no work", a, b, planes, 1, L);
    run<1024, 16, 3, 4, 0, 0, true, 4>("plane model: sliced loads, no work", a, b, planes, 1, L);
    run<1024, 16, 3, 4, 0, 8, true, 4>("plane model: sliced loads, work 8", a, b, planes, 1, L);
    run<1024, 16, 3, 4, 0, 8, false, 4>("plane model: ... wave-private trips", a, b, planes, 1, L);
    run<1024, 16, 3, 3, 1, 8, true, 4>("deferred stores: 1024 thr, barriers", a, b, planes, 1, L);
    run<1024, 16, 3, 3, 1, 8, false, 4>("deferred stores: 1024 thr, wave-private", a, b, planes, 1, L);
    run<1024, 16, 3, 2, 1, 8, false, 4>("deferred stores: 1024 thr, wp, 2 load slices", a, b, planes, 1, L);
    run<512, 32, 3, 4, 0, 8, false, 2>("plane model 512 thr: sliced loads, wp", a, b, planes, 1, L);
    run<512, 32, 3, 3, 1, 8, false, 2>("deferred stores: 512 thr, wp", a, b, planes, 1, L);
    run<512, 32, 3, 3, 1, 8, true, 2>("deferred stores: 512 thr, barriers", a, b, planes, 1, L);
    run<512, 32, 3, 2, 1, 8, false, 2>("deferred stores: 512 thr, wp, 2 load slices", a, b, planes, 1, L);
    // what two / four smaller workgroups per CU would buy (half / quarter planes: not an FFT option, a bound)
    run<512, 16, 3, 1, 0, 8, true, 4>("half planes, 2 wg/cu, burst, barriers", a, b, planes * 2, 2, 68 * 1024);
    run<256, 16, 3, 1, 0, 8, true, 4>("quarter planes, 4 wg/cu, burst, barriers", a, b, planes * 4, 4, 34 * 1024);
    run<1024, 16, 3, 4, 0, 16, true, 4>("plane model: sliced loads, work 16", a, b, planes, 1, L);
    run<1024, 16, 3, 3, 1, 16, false, 4>("deferred stores: 1024 thr, wp, work 16", a, b, planes, 1, L);
    // ---- 100 x 640 x 480: 16-column x 640-row tiles in place (the second pass of BASELINE config 4), 87 KB of LDS ----
    {
        const long long e4 = 100LL * 640 * 480;
        f2* c;
        CK(hipMalloc(&c, e4 * 8));
        CK(hipMemset(c, 0, e4 * 8));
        const long long tiles = 100LL * 480 / 16;
        const int L4 = 87 * 1024;
        run<640, 16, 3, 1, 0, 0, true, 1, 16>("cols640 model: burst, no work", c, c, tiles, 1, L4, 480);
        run<640, 16, 3, 4, 0, 8, true, 1, 16>("cols640 model: sliced loads, work 8", c, c, tiles, 1, L4, 480);
        run<640, 16, 3, 4, 0, 8, false, 1, 16>("cols640 model: sliced loads, work 8, wp", c, c, tiles, 1, L4, 480);
        run<640, 16, 3, 3, 1, 8, true, 1, 16>("cols640 model: deferred stores, barriers", c, c, tiles, 1, L4, 480);
        run<640, 16, 3, 3, 1, 8, false, 1, 16>("cols640 model: deferred stores, wp", c, c, tiles, 1, L4, 480);
        run<640, 16, 3, 2, 1, 8, false, 1, 16>("cols640 model: deferred stores, wp, 2 load slices", c, c, tiles, 1, L4, 480);
        run<320, 16, 3, 1, 0, 8, true, 1, 8>("cols640 8-col tiles, 3 wg/cu, burst", c, c, tiles * 2, 3, 44 * 1024, 480);
        run<640, 8, 3, 1, 0, 8, true, 1, 16>("cols320 (half-height) tiles, 3 wg/cu, burst", c, c, tiles * 2, 3, 44 * 1024, 480);
        run<256, 5, 3, 1, 0, 8, true, 1, 16>("cols80 16-col tiles (10 KB), 8 wg/cu, burst", c, c, tiles * 8, 8, 12 * 1024, 480);
        run<512, 5, 3, 1, 0, 8, true, 1, 32>("cols80 32-col tiles (20 KB), 6 wg/cu, burst", c, c, tiles * 4, 6, 22 * 1024, 480);
    }
    return 0;
}
