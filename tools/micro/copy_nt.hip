// copy_nt.hip -- can the ~5.6 TB/s row-shaped copy ceiling be raised?  Non-temporal loads/stores,
// persistent grid sizes, wave-per-row vs workgroup-per-4-rows.  Design evidence only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void copy_row(const f2* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f2* p = in + (size_t)r * 1024 + lane;
        f2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NT_LD ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (NT_ST) __builtin_nontemporal_store(v[j], q + j * 64); else q[j * 64] = v[j];
        }
    }
}
// two rows in flight per wave (software pipelined)
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void copy_row_pf(const f2* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    f2 nx[16];
    int r = wave;
    if (r < rows) {
        const f2* p = in + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = NT_LD ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
    }
    for (; r < rows; r += nw) {
        f2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nx[j];
        if (r + nw < rows) {
            const f2* p = in + (size_t)(r + nw) * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) nx[j] = NT_LD ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        }
        f2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (NT_ST) __builtin_nontemporal_store(v[j], q + j * 64); else q[j * 64] = v[j];
        }
    }
}

int main() {
    const size_t bytes = 100000ull * 1024 * 8;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, int blocks, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            const int it = 30;
            for (int i = 0; i < it; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
            if (ms < best) best = ms;
        }
        printf("%-26s grid %5d  %8.4f ms  %7.1f GB/s\n", name, blocks, best, 2.0 * bytes / best / 1e6);
    };
    for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 4096, 25000}) {
        time("plain", blocks, [&] { copy_row<0, 0><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
        time("nt load", blocks, [&] { copy_row<1, 0><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
        time("nt store", blocks, [&] { copy_row<0, 1><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
        time("nt both", blocks, [&] { copy_row<1, 1><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
        time("pf plain", blocks, [&] { copy_row_pf<0, 0><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
        time("pf nt both", blocks, [&] { copy_row_pf<1, 1><<<blocks, 256>>>((f2*)a, (f2*)b, 100000); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
