// rw_mix.hip -- what a kernel with the traffic mix of a REAL-input transform (4 B read + 8 B written per point) can
// reach: read-only, write-only, promote (float in, float2 out) with 4 / 8 / 16 bytes per lane on the load side, against
// the 8 B + 8 B copy.  100k rows x 1024 points (the reference's rfft benchmark shape).  Design evidence only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void write_only(f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        f2* q = out + (size_t)r * 1024 + lane;
        f2 v = {(float)r, (float)lane};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (NT) __builtin_nontemporal_store(v, q + j * 64); else q[j * 64] = v;
        }
    }
}
template <int NT>
__global__ __launch_bounds__(256) void read_only(const f2* __restrict__ in, f2* __restrict__ sink, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    f2 acc = {0.f, 0.f};
    for (int r = wave; r < rows; r += nw) {
        const f2* p = in + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}
// promote: 4 B per lane loads (what the pass-0 load of a real-input row kernel does), 8 B per lane stores
template <int NT>
__global__ __launch_bounds__(256) void promote4(const float* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float* p = in + (size_t)r * 1024 + lane;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f2 w = {v[j], 0.f};
            if (NT) __builtin_nontemporal_store(w, q + j * 64); else q[j * 64] = w;
        }
    }
}
// promote with 16 B per lane loads (4 consecutive reals), stores of 8 B per lane after a register shuffle-free layout:
// lane l holds points 4l..4l+3 of a 256-point chunk and writes them as two 16-B stores
template <int NT>
__global__ __launch_bounds__(256) void promote16(const float* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f4* p = (const f4*)(in + (size_t)r * 1024) + lane;
        f4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f4* q = (f4*)(out + (size_t)r * 1024) + 2 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4 a = {v[j].x, 0.f, v[j].y, 0.f}, b = {v[j].z, 0.f, v[j].w, 0.f};
            if (NT) { __builtin_nontemporal_store(a, q + j * 128); __builtin_nontemporal_store(b, q + j * 128 + 1); }
            else { q[j * 128] = a; q[j * 128 + 1] = b; }
        }
    }
}
template <int NT>
__global__ __launch_bounds__(256) void copy8(const f2* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f2* p = in + (size_t)r * 1024 + lane;
        f2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (NT) __builtin_nontemporal_store(v[j], q + j * 64); else q[j * 64] = v[j];
        }
    }
}

// 16 B per lane copy: a wave moves 1 KB per instruction (row-shaped, 8 instructions per 8-KB row)
template <int NT>
__global__ __launch_bounds__(256) void copy16(const f4* __restrict__ in, f4* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f4* p = in + (size_t)r * 512 + lane;
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f4* q = out + (size_t)r * 512 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (NT) __builtin_nontemporal_store(v[j], q + j * 64); else q[j * 64] = v[j];
        }
    }
}
// 16 B loads, 8 B stores and the reverse: which side is it?
template <int NT>
__global__ __launch_bounds__(256) void copy16_8(const f4* __restrict__ in, f2* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f4* p = in + (size_t)r * 512 + lane;
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f2* q = out + (size_t)r * 1024 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // (permuted inside the row: timing only)
            f2 a = {v[j].x, v[j].y}, b = {v[j].z, v[j].w};
            if (NT) { __builtin_nontemporal_store(a, q + (2 * j) * 64); __builtin_nontemporal_store(b, q + (2 * j + 1) * 64); }
            else { q[(2 * j) * 64] = a; q[(2 * j + 1) * 64] = b; }
        }
    }
}
template <int NT>
__global__ __launch_bounds__(256) void copy8_16(const f2* __restrict__ in, f4* __restrict__ out, int rows) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const f2* p = in + (size_t)r * 1024 + lane;
        f2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
        f4* q = out + (size_t)r * 512 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f4 a = {v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y};
            if (NT) __builtin_nontemporal_store(a, q + j * 64); else q[j * 64] = a;
        }
    }
}

int main() {
    const int rows = 100000;
    const size_t bytes = (size_t)rows * 1024 * 8;
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double moved, auto launch) {
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
        printf("%-28s %8.4f ms  %7.1f GB/s\n", name, best, moved / best / 1e6);
    };
    for (int blocks : {1024, 2048, 4096}) {
        printf("grid %d\n", blocks);
        time("read-only 8 B/lane", bytes, [&] { read_only<0><<<blocks, 256>>>((f2*)a, (f2*)b, rows); });
        time("read-only nt", bytes, [&] { read_only<1><<<blocks, 256>>>((f2*)a, (f2*)b, rows); });
        time("write-only 8 B/lane", bytes, [&] { write_only<0><<<blocks, 256>>>((f2*)b, rows); });
        time("write-only nt", bytes, [&] { write_only<1><<<blocks, 256>>>((f2*)b, rows); });
        time("copy 8+8", 2.0 * bytes, [&] { copy8<0><<<blocks, 256>>>((f2*)a, (f2*)b, rows); });
        time("copy 8+8 nt", 2.0 * bytes, [&] { copy8<1><<<blocks, 256>>>((f2*)a, (f2*)b, rows); });
        time("copy 16+16", 2.0 * bytes, [&] { copy16<0><<<blocks, 256>>>((f4*)a, (f4*)b, rows); });
        time("copy 16+16 nt", 2.0 * bytes, [&] { copy16<1><<<blocks, 256>>>((f4*)a, (f4*)b, rows); });
        time("copy 16 ld + 8 st nt", 2.0 * bytes, [&] { copy16_8<1><<<blocks, 256>>>((f4*)a, (f2*)b, rows); });
        time("copy 8 ld + 16 st nt", 2.0 * bytes, [&] { copy8_16<1><<<blocks, 256>>>((f2*)a, (f4*)b, rows); });
        time("promote 4 B loads", 1.5 * bytes, [&] { promote4<0><<<blocks, 256>>>((float*)a, (f2*)b, rows); });
        time("promote 4 B loads nt", 1.5 * bytes, [&] { promote4<1><<<blocks, 256>>>((float*)a, (f2*)b, rows); });
        time("promote 16 B loads", 1.5 * bytes, [&] { promote16<0><<<blocks, 256>>>((float*)a, (f2*)b, rows); });
        time("promote 16 B loads nt", 1.5 * bytes, [&] { promote16<1><<<blocks, 256>>>((float*)a, (f2*)b, rows); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
