// lane_chunks.hip -- how fast are HBM accesses in which every LANE owns a contiguous chunk of CH elements (8 bytes
// each) and a wave instruction touches 64 different chunks (stride CH * 8 bytes across lanes)?  This is the access shape
// of a register-resident 93-point kernel (radix 31 in registers, radix 3 across 3 adjacent lanes by DPP): a lane ends
// with 31 consecutive outputs.  Compared with the coalesced shape (lanes along consecutive elements).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o lane_chunks lane_chunks.hip && ./lane_chunks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: load coalesced, store chunked; 1: load chunked, store coalesced; 2: both chunked; 3: both coalesced
template <int CH, int MODE, int NT>
__global__ __launch_bounds__(256) void k(const f2* __restrict__ in, f2* __restrict__ out, long long n_waves_total) {
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave; w < n_waves_total; w += nw) {
        const f2* pi = in + w * 64 * CH;
        f2* po = out + w * 64 * CH;
        f2 v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const f2* a = (MODE == 1 || MODE == 2) ? pi + lane * CH + j : pi + j * 64 + lane;
            v[j] = NT ? __builtin_nontemporal_load(a) : *a;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            f2* a = (MODE == 0 || MODE == 2) ? po + lane * CH + j : po + j * 64 + lane;
            if (NT) __builtin_nontemporal_store(v[j], a); else *a = v[j];
        }
    }
}

template <int CH, int MODE, int NT>
int run(const char* name, f2* a, f2* b, size_t elems) {
    const long long waves = (long long)(elems / (64 * CH));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int grid : {1024, 2048, 4096}) {
        for (int i = 0; i < 3; ++i) k<CH, MODE, NT><<<grid, 256>>>(a, b, waves);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) k<CH, MODE, NT><<<grid, 256>>>(a, b, waves);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        if (ms < best) best = ms;
    }
    const double bytes = (double)waves * 64 * CH * 8;
    printf("%-44s CH %2d nt %d: %7.4f ms %7.1f GB/s\n", name, CH, NT, best, 2.0 * bytes / best / 1e6);
    return 0;
}

int main() {
    const size_t elems = 46500000;  // 372 MB, the 500k x 93 tensor
    f2 *a, *b;
    CK(hipMalloc(&a, elems * 8));
    CK(hipMalloc(&b, elems * 8));
    CK(hipMemset(a, 0, elems * 8));
    CK(hipMemset(b, 0, elems * 8));
    run<31, 3, 0>("both coalesced", a, b, elems);
    run<31, 3, 1>("both coalesced", a, b, elems);
    run<31, 0, 0>("load coalesced, store 31-element chunks", a, b, elems);
    run<31, 0, 1>("load coalesced, store 31-element chunks", a, b, elems);
    run<31, 1, 0>("load 31-element chunks, store coalesced", a, b, elems);
    run<31, 1, 1>("load 31-element chunks, store coalesced", a, b, elems);
    run<31, 2, 0>("both 31-element chunks", a, b, elems);
    run<31, 2, 1>("both 31-element chunks", a, b, elems);
    run<16, 2, 0>("both 16-element chunks", a, b, elems);
    run<8, 2, 0>("both 8-element chunks", a, b, elems);
    run<3, 2, 0>("both 3-element chunks (the pass-0 loads)", a, b, elems);
    return 0;
}
