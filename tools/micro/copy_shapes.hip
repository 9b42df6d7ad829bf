// copy_shapes.hip -- HBM streaming microbenchmark in the access shapes an FFT row kernel can use.
// Not part of the product; design evidence only (DESIGN.md "access width").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// A/B: flat grid-stride copies
template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void copy_flat(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) out[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// C: wave per 8 KiB row, float2 per lane, 16 loads at stride 64 complex
__global__ __launch_bounds__(256) void copy_row_f2(const float2* __restrict__ in, float2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float2* p = in + (size_t)r * 1024 + lane;
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[j * 64];
        float2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j * 64] = v[j];
    }
}
// D: wave per row, float4 per lane, 8 loads at stride 128 complex
__global__ __launch_bounds__(256) void copy_row_f4(const float4* __restrict__ in, float4* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float4* p = in + (size_t)r * 512 + lane;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j * 64];
        float4* q = out + (size_t)r * 512 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j * 64] = v[j];
    }
}
// E: 8 lanes per 1 KiB row (N=128), float4 per lane, 8 loads at stride 16 complex; 8 rows per wave
__global__ __launch_bounds__(256) void copy_row128_f4(const float4* __restrict__ in, float4* __restrict__ out, int rows) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int grp = tid >> 3, l = tid & 7, ng = (gridDim.x * blockDim.x) >> 3;
    for (int r = grp; r < rows; r += ng) {
        const float4* p = in + (size_t)r * 64 + l;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j * 8];
        float4* q = out + (size_t)r * 64 + l;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j * 8] = v[j];
    }
}
// F: 16 lanes per 1 KiB row, float2 per lane, 8 loads at stride 16 complex; 4 rows per wave
__global__ __launch_bounds__(256) void copy_row128_f2(const float2* __restrict__ in, float2* __restrict__ out, int rows) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int grp = tid >> 4, l = tid & 15, ng = (gridDim.x * blockDim.x) >> 4;
    for (int r = grp; r < rows; r += ng) {
        const float2* p = in + (size_t)r * 128 + l;
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j * 16];
        float2* q = out + (size_t)r * 128 + l;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j * 16] = v[j];
    }
}

int main() {
    const size_t bytes = 100000ull * 1024 * 8;  // config 2 tensor
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0, 0);
        const int it = 50;
        for (int i = 0; i < it; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        printf("%-28s %8.4f ms  %7.1f GB/s (R+W)\n", name, ms, 2.0 * bytes / ms / 1e6);
    };
    for (int blocks : {1024, 2048, 4096, 8192}) {
        printf("grid %d x 256\n", blocks);
        time("flat float4 x4", [&] { copy_flat<float4, 4><<<blocks, 256>>>((float4*)a, (float4*)b, bytes / 16); });
        time("flat float2 x4", [&] { copy_flat<float2, 4><<<blocks, 256>>>((float2*)a, (float2*)b, bytes / 8); });
        time("flat float2 x8", [&] { copy_flat<float2, 8><<<blocks, 256>>>((float2*)a, (float2*)b, bytes / 8); });
        time("row1024 float2 x16", [&] { copy_row_f2<<<blocks, 256>>>((float2*)a, (float2*)b, 100000); });
        time("row1024 float4 x8", [&] { copy_row_f4<<<blocks, 256>>>((float4*)a, (float4*)b, 100000); });
        time("row128 8lanes float4 x8", [&] { copy_row128_f4<<<blocks, 256>>>((float4*)a, (float4*)b, 800000); });
        time("row128 16lanes float2 x8", [&] { copy_row128_f2<<<blocks, 256>>>((float2*)a, (float2*)b, 800000); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
