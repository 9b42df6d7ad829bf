// copy_probe.hip -- libmifft_probe.so: COPY kernels with the tile shapes of the FFT passes, for bench.py's
// `roofline.copy` (what this chip gives a kernel that only moves the bytes of a pass, measured in the same run on the
// same box as the FFT).  Measurement tooling: not part of libmifft.so, never loaded by the product package.
//
//   probe_copy_flat   a contiguous tensor moved once, out of place (the row passes and the fused planes: whole rows /
//                     planes are contiguous runs), persistent workgroups, next chunk prefetched into registers,
//                     optional non-temporal loads / stores, optional LDS reservation so that the occupancy equals the
//                     FFT kernel's (a 136-KB plane leaves one workgroup per CU)
//   probe_copy_cols   tiles of W adjacent columns x N rows of a [outer][N][inner] tensor (runs of W * 8 bytes at a row
//                     pitch of inner * 8 bytes), in place or out of place, one LDS round trip, the FFT tile's LDS
//                     footprint
// Byte convention as the reference's cuFFT harness (cufft-benchmark-main/cufft_benchmark.cu:52-53): one complex read + one
// complex write per element.
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int NT_LD>
__device__ __forceinline__ f2 ld(const f2* p) {
    if constexpr (NT_LD) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT_ST>
__device__ __forceinline__ void st(f2* p, f2 v) {
    if constexpr (NT_ST) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// chunk = THREADS * E consecutive elements; lanes take consecutive elements (8 B per lane, 512-B wave runs)
template <int THREADS, int E, int NT_LD, int NT_ST, bool LDS_TRIP>
__global__ __launch_bounds__(THREADS) void copy_flat(const f2* __restrict__ in, f2* __restrict__ out, long long n_elems) {
    extern __shared__ f2 lds[];
    constexpr long long CHUNK = (long long)THREADS * E;
    const long long n_chunks = (n_elems + CHUNK - 1) / CHUNK;
    const int tid = threadIdx.x;
    f2 nx[E];
    long long t = blockIdx.x;
    auto load = [&](long long c, f2* v) {
        const long long b = c * CHUNK + tid;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            long long i = b + (long long)e * THREADS;
            if (i >= n_elems) i = n_elems - 1;  // ragged last chunk: re-read a valid element, never stored
            v[e] = ld<NT_LD>(in + i);
        }
    };
    if (t < n_chunks) load(t, nx);
    for (; t < n_chunks; t += gridDim.x) {
        f2 v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = nx[e];
        if (t + gridDim.x < n_chunks) load(t + gridDim.x, nx);
        if constexpr (LDS_TRIP) {
#pragma unroll
            for (int e = 0; e < E; ++e) lds[e * THREADS + tid] = v[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = lds[e * THREADS + (tid ^ 1)];
            __syncthreads();
        }
        const long long b = t * CHUNK + tid;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const long long i = b + (long long)e * THREADS;
            if (i < n_elems) st<NT_ST>(out + i, v[e]);
        }
    }
}

template <int N, int W, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_cols(const f2* __restrict__ in, f2* __restrict__ out, long long n_tiles,
                                                     int inner, int tiles_per_outer) {
    extern __shared__ f2 lds[];
    static_assert((N * W) % THREADS == 0 && THREADS % W == 0, "whole sweeps");
    constexpr int E = N * W / THREADS, ROWS_PER_IT = THREADS / W;
    const int tid = threadIdx.x, c = tid % W, r0 = tid / W;
    f2 v[E], nx[E];
    auto base_of = [&](long long t) {
        const long long o = t / tiles_per_outer;
        return o * (long long)N * inner + (t - o * tiles_per_outer) * W;
    };
    long long t = blockIdx.x;
    if (t < n_tiles) {
        const f2* g = in + base_of(t) + (long long)r0 * inner + c;
#pragma unroll
        for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner];
    }
    for (; t < n_tiles; t += gridDim.x) {
        const long long b = base_of(t);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = nx[e];
        if (t + gridDim.x < n_tiles) {
            const f2* g = in + base_of(t + gridDim.x) + (long long)r0 * inner + c;
#pragma unroll
            for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) lds[(e * ROWS_PER_IT + r0) * W + c] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = lds[(e * ROWS_PER_IT + r0) * W + (c ^ 1)];
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0x0F70);  // the prefetched tile has landed before the stores queue behind it (vmcnt is in order)
        f2* g = out + b + (long long)r0 * inner + c;
#pragma unroll
        for (int e = 0; e < E; ++e) g[(long long)e * ROWS_PER_IT * inner] = v[e];
    }
}

static int time_launches(hipStream_t s, int iters, float* ms_out, void (*launch)(void*, hipStream_t), void* arg) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    for (int i = 0; i < 3; ++i) launch(arg, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch(arg, s);
    hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return -1;
    *ms_out = ms / (float)iters;
    return 0;
}

struct FlatArgs {
    const f2* in;
    f2* out;
    long long n;
    int nt, lds_bytes, wg_per_cu, cus;
};
struct ColsArgs {
    const f2* in;
    f2* out;
    long long outer;
    int n, inner, w, wg_per_cu, cus;
};

template <int THREADS, int E, int NTL, int NTS, bool TRIP>
static void launch_flat_t(const FlatArgs& a, hipStream_t s) {
    auto k = copy_flat<THREADS, E, NTL, NTS, TRIP>;
    int lds = a.lds_bytes;
    if (TRIP && lds < (int)(THREADS * E * sizeof(f2))) lds = THREADS * E * sizeof(f2);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long chunks = (a.n + (long long)THREADS * E - 1) / ((long long)THREADS * E);
    long long grid = (long long)a.cus * a.wg_per_cu;
    if (grid > chunks) grid = chunks;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(THREADS), lds, s, a.in, a.out, a.n);
}
static void launch_flat(void* p, hipStream_t s) {
    const FlatArgs& a = *(const FlatArgs*)p;
    const bool trip = a.lds_bytes > 0;
    // 256 threads x 16 elements = 32 KiB per workgroup chunk; with an LDS reservation the workgroup is 1024 threads like the plane
    if (trip) {
        if (a.nt & 1) launch_flat_t<1024, 16, 1, 0, true>(a, s);
        else launch_flat_t<1024, 16, 0, 0, true>(a, s);
    } else {
        switch (a.nt & 3) {
            case 0: launch_flat_t<256, 16, 0, 0, false>(a, s); break;
            case 1: launch_flat_t<256, 16, 1, 0, false>(a, s); break;
            case 2: launch_flat_t<256, 16, 0, 1, false>(a, s); break;
            default: launch_flat_t<256, 16, 1, 1, false>(a, s); break;
        }
    }
}

template <int N, int W, int THREADS>
static void launch_cols_t(const ColsArgs& a, hipStream_t s) {
    auto k = copy_cols<N, W, THREADS>;
    const int lds = N * W * (int)sizeof(f2);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tpo = a.inner / W;
    const long long n_tiles = a.outer * tpo;
    long long grid = (long long)a.cus * a.wg_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(THREADS), lds, s, a.in, a.out, n_tiles, a.inner, tpo);
}
static void launch_cols(void* p, hipStream_t s) {
    const ColsArgs& a = *(const ColsArgs*)p;
    if (a.n == 640 && a.w == 16) launch_cols_t<640, 16, 640>(a, s);
    else if (a.n == 480 && a.w == 16) launch_cols_t<480, 16, 640>(a, s);
    else if (a.n == 128 && a.w == 32) launch_cols_t<128, 32, 512>(a, s);
    else if (a.n == 128 && a.w == 16) launch_cols_t<128, 16, 256>(a, s);
    else if (a.n == 64 && a.w == 64) launch_cols_t<64, 64, 512>(a, s);
    else if (a.n == 256 && a.w == 16) launch_cols_t<256, 16, 256>(a, s);
}

static int num_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
}

extern "C" {

// nt: bit 0 non-temporal loads, bit 1 non-temporal stores.  lds_bytes > 0: reserve that much LDS per workgroup (and make
// one LDS round trip), 1024-thread workgroups.  Returns 0 and the average ms per copy of n_elems complex64 elements.
int probe_copy_flat(const void* in, void* out, long long n_elems, int nt, int lds_bytes, int wg_per_cu, int iters,
                    void* stream, float* ms_out) {
    if (!in || !out || !ms_out || n_elems < 1 || iters < 1 || lds_bytes > 160 * 1024) return -1;
    FlatArgs a{(const f2*)in, (f2*)out, n_elems, nt, lds_bytes, wg_per_cu > 0 ? wg_per_cu : 8, num_cus()};
    return time_launches((hipStream_t)stream, iters, ms_out, launch_flat, &a);
}

// column tiles: (n, w) one of (640,16) (480,16) (128,32) (128,16) (64,64) (256,16); inner % w == 0; in == out: in place
int probe_copy_cols(const void* in, void* out, long long outer, int n, int inner, int w, int wg_per_cu, int iters,
                    void* stream, float* ms_out) {
    if (!in || !out || !ms_out || outer < 1 || iters < 1 || inner % w != 0) return -1;
    const bool known = (n == 640 && w == 16) || (n == 480 && w == 16) || (n == 128 && (w == 32 || w == 16)) ||
                       (n == 64 && w == 64) || (n == 256 && w == 16);
    if (!known) return -2;
    ColsArgs a{(const f2*)in, (f2*)out, outer, n, inner, w, wg_per_cu > 0 ? wg_per_cu : 1, num_cus()};
    return time_launches((hipStream_t)stream, iters, ms_out, launch_cols, &a);
}

// The two passes of an N-D transform as copies, ALTERNATING like the transform does (pass 1 out of place x -> out as a flat
// copy, pass 2 in place on out as column tiles): what the second pass finds in the caches depends on the first, so the pair
// is timed as a pair.  ms_out[0] = average per pair; per-pass figures come from the separate entry points above.
int probe_copy_pair(const void* x, void* out, long long n_elems, int nt1, int lds1, int wg1, long long outer, int n, int inner,
                    int w, int wg2, int iters, void* stream, float* ms_out) {
    if (!x || !out || !ms_out || n_elems < 1 || iters < 1 || inner % w != 0 || lds1 > 160 * 1024) return -1;
    FlatArgs a{(const f2*)x, (f2*)out, n_elems, nt1, lds1, wg1 > 0 ? wg1 : 8, num_cus()};
    ColsArgs c{(const f2*)out, (f2*)out, outer, n, inner, w, wg2 > 0 ? wg2 : 1, num_cus()};
    struct Pair {
        FlatArgs* a;
        ColsArgs* c;
    } pr{&a, &c};
    return time_launches((hipStream_t)stream, iters, ms_out,
                         [](void* p, hipStream_t s) {
                             Pair* q = (Pair*)p;
                             launch_flat(q->a, s);
                             launch_cols(q->c, s);
                         },
                         &pr);
}

}  // extern "C"
