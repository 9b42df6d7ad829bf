// Ceiling of the column-tile access pattern: a workgroup moves a tile of W adjacent columns x N rows of a
// [outer][N][inner] float2 tensor (runs of W*8 bytes, row pitch inner*8 bytes) through registers, in place or
// out of place, with the LDS footprint of the real kernel reserved so that the occupancy matches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o copy_cols copy_cols.hip && ./copy_cols
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int N, int W, int THREADS, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void copy_cols(const float2* __restrict__ in, float2* __restrict__ out,
                                                     long long n_tiles, int inner, int tiles_per_outer) {
    extern __shared__ float2 lds[];
    constexpr int E = N * W / THREADS;       // elements per thread
    constexpr int ROWS_PER_IT = THREADS / W;  // rows covered by one sweep of the workgroup
    const int tid = threadIdx.x;
    const int c = tid % W, r0 = tid / W;
    float2 v[E], nx[E];
    auto base_of = [&](long long t) {
        const long long o = t / tiles_per_outer;
        const long long c0 = (t - o * tiles_per_outer) * W;
        return o * (long long)N * inner + c0;
    };
    long long t = blockIdx.x;
    if (PREFETCH && t < n_tiles) {
        const float2* g = in + base_of(t) + (long long)r0 * inner + c;
#pragma unroll
        for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner];
    }
    for (; t < n_tiles; t += gridDim.x) {
        const long long b = base_of(t);
        if (PREFETCH) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = nx[e];
            const long long tn = t + gridDim.x;
            if (tn < n_tiles) {
                const float2* g = in + base_of(tn) + (long long)r0 * inner + c;
#pragma unroll
                for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner];
            }
        } else {
            const float2* g = in + b + (long long)r0 * inner + c;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = g[(long long)e * ROWS_PER_IT * inner];
        }
        // one LDS round trip so that the data really passes through the workgroup like the FFT does
#pragma unroll
        for (int e = 0; e < E; ++e) lds[(e * ROWS_PER_IT + r0) * W + c] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = lds[(e * ROWS_PER_IT + r0) * W + (c ^ 1)];
        __syncthreads();
        float2* g = out + b + (long long)r0 * inner + c;
#pragma unroll
        for (int e = 0; e < E; ++e) g[(long long)e * ROWS_PER_IT * inner] = v[e];
    }
}

// the same tile with 16 bytes per lane: a lane moves TWO adjacent columns (global_load_dwordx4 / ds_*_b128), half the
// vector-memory and LDS instructions of copy_cols for the same bytes
template <int N, int W, int THREADS, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void copy_cols_v4(const float4* __restrict__ in, float4* __restrict__ out,
                                                        long long n_tiles, int inner, int tiles_per_outer) {
    extern __shared__ float4 lds4[];
    constexpr int W2 = W / 2;
    constexpr int E = N * W2 / THREADS;        // float4 elements per thread
    constexpr int ROWS_PER_IT = THREADS / W2;  // rows covered by one sweep of the workgroup
    const int tid = threadIdx.x;
    const int c = tid % W2, r0 = tid / W2;
    const int inner2 = inner / 2;
    float4 v[E], nx[E];
    auto base_of = [&](long long t) {
        const long long o = t / tiles_per_outer;
        const long long c0 = (t - o * tiles_per_outer) * W2;
        return o * (long long)N * inner2 + c0;
    };
    long long t = blockIdx.x;
    if (PREFETCH && t < n_tiles) {
        const float4* g = in + base_of(t) + (long long)r0 * inner2 + c;
#pragma unroll
        for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner2];
    }
    for (; t < n_tiles; t += gridDim.x) {
        const long long b = base_of(t);
        if (PREFETCH) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = nx[e];
            const long long tn = t + gridDim.x;
            if (tn < n_tiles) {
                const float4* g = in + base_of(tn) + (long long)r0 * inner2 + c;
#pragma unroll
                for (int e = 0; e < E; ++e) nx[e] = g[(long long)e * ROWS_PER_IT * inner2];
            }
        } else {
            const float4* g = in + b + (long long)r0 * inner2 + c;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = g[(long long)e * ROWS_PER_IT * inner2];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) lds4[(e * ROWS_PER_IT + r0) * W2 + c] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = lds4[(e * ROWS_PER_IT + r0) * W2 + (c ^ 1)];
        __syncthreads();
        float4* g = out + b + (long long)r0 * inner2 + c;
#pragma unroll
        for (int e = 0; e < E; ++e) g[(long long)e * ROWS_PER_IT * inner2] = v[e];
    }
}

template <int N, int W, int THREADS, bool PF>
int run4(const char* name, long long outer, int inner, bool inplace, int wg_per_cu, float2* a, float2* b) {
    const int tpo = inner / W;
    const long long n_tiles = outer * tpo;
    const size_t lds = (size_t)N * W * sizeof(float2);
    CK(hipFuncSetAttribute((const void*)copy_cols_v4<N, W, THREADS, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = 256 * wg_per_cu;
    if (grid > n_tiles) grid = (int)n_tiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float2* dst = inplace ? a : b;
    for (int i = 0; i < 5; ++i) copy_cols_v4<N, W, THREADS, PF><<<grid, THREADS, lds>>>((const float4*)a, (float4*)dst, n_tiles, inner, tpo);
    CK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) copy_cols_v4<N, W, THREADS, PF><<<grid, THREADS, lds>>>((const float4*)a, (float4*)dst, n_tiles, inner, tpo);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)outer * N * inner * 8.0;
    printf("%-44s N %4d W %2d thr %4d pf %d %s wg/cu %d lds %6zu: %7.4f ms %7.1f GB/s  (16 B per lane)\n", name, N, W, THREADS,
           (int)PF, inplace ? "inplace" : "out    ", wg_per_cu, lds, ms, 2.0 * bytes / ms / 1e6);
    return 0;
}

template <int N, int W, int THREADS, bool PF>
int run(const char* name, long long outer, int inner, bool inplace, int wg_per_cu, float2* a, float2* b) {
    const int tpo = inner / W;
    const long long n_tiles = outer * tpo;
    const size_t lds = (size_t)N * W * sizeof(float2);
    CK(hipFuncSetAttribute((const void*)copy_cols<N, W, THREADS, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = 256 * wg_per_cu;
    if (grid > n_tiles) grid = (int)n_tiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float2* dst = inplace ? a : b;
    for (int i = 0; i < 5; ++i) copy_cols<N, W, THREADS, PF><<<grid, THREADS, lds>>>(a, dst, n_tiles, inner, tpo);
    CK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) copy_cols<N, W, THREADS, PF><<<grid, THREADS, lds>>>(a, dst, n_tiles, inner, tpo);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)outer * N * inner * 8.0;
    printf("%-44s N %4d W %2d thr %4d pf %d %s wg/cu %d lds %6zu: %7.4f ms %7.1f GB/s\n", name, N, W, THREADS, (int)PF,
           inplace ? "inplace" : "out    ", wg_per_cu, lds, ms, 2.0 * bytes / ms / 1e6);
    return 0;
}

int main() {
    const size_t max_elems = (size_t)64 << 20;  // 512 MiB per buffer
    float2 *a, *b;
    CK(hipMalloc(&a, max_elems * 8));
    CK(hipMalloc(&b, max_elems * 8));
    CK(hipMemset(a, 0, max_elems * 8));
    CK(hipMemset(b, 0, max_elems * 8));
    // four-step pass: 64 transforms of 2^20 as [64][1024][1024]
    for (int ip = 0; ip < 2; ++ip) {
        run<1024, 16, 512, false>("fourstep 1024x1024", 64, 1024, ip, 1, a, b);
        run<1024, 16, 1024, false>("fourstep 1024x1024", 64, 1024, ip, 1, a, b);
        run<1024, 16, 512, true>("fourstep 1024x1024", 64, 1024, ip, 1, a, b);
        run<1024, 16, 1024, true>("fourstep 1024x1024", 64, 1024, ip, 1, a, b);
        run<1024, 8, 512, false>("fourstep 1024x1024", 64, 1024, ip, 2, a, b);
        run<1024, 8, 256, false>("fourstep 1024x1024", 64, 1024, ip, 2, a, b);
        run<1024, 8, 512, true>("fourstep 1024x1024", 64, 1024, ip, 2, a, b);
        run<1024, 4, 256, false>("fourstep 1024x1024", 64, 1024, ip, 4, a, b);
        run<256, 32, 512, false>("256-row tiles, 32 cols", 256, 1024, ip, 2, a, b);
        run<256, 64, 512, false>("256-row tiles, 64 cols", 256, 1024, ip, 1, a, b);
        run<256, 16, 256, false>("256-row tiles, 16 cols", 256, 1024, ip, 4, a, b);
        run<128, 16, 256, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 8, a, b);
        run<128, 32, 256, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 4, a, b);
        run<128, 64, 512, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 2, a, b);
        run4<640, 16, 512, false>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run4<640, 16, 512, true>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run4<640, 16, 256, false>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run4<640, 32, 512, false>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run4<128, 16, 256, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 8, a, b);
        run4<128, 32, 256, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 4, a, b);
        run4<128, 64, 512, false>("128^3 dim 0 (inner 16384)", 10, 16384, ip, 2, a, b);
        run4<1024, 16, 512, false>("fourstep 1024x1024", 64, 1024, ip, 1, a, b);
        run<640, 16, 512, false>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run<640, 16, 512, true>("100x640x480 cols", 100, 480, ip, 1, a, b);
        run<640, 8, 256, false>("100x640x480 cols", 100, 480, ip, 3, a, b);
        run<640, 32, 1024, false>("100x640x480 cols", 100, 480, ip, 1, a, b);
    }
    return 0;
}
