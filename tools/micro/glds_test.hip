// glds_test.hip -- semantics check of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950:
// per-lane global source, LDS destination = wave-uniform base + lane*16; EXEC-masked lanes; vmcnt wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__global__ __launch_bounds__(256) void k(const float4* __restrict__ in, float4* __restrict__ out, int n16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* lds = (float4*)smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // each wave copies pieces wave, wave+4, ... of 64 float4 each
    for (int piece = wave; piece * 64 < n16; piece += 4) {
        const int idx = piece * 64 + lane;
        if (idx < n16) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(in + idx), (lds_ptr_t)(lds + piece * 64), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < n16; i += 256) {
        float4 v = lds[i];
        v.x += 1.0f;
        out[i] = v;
    }
}

int main() {
    const int n16 = 1000;  // not a multiple of 64: last piece partially masked
    std::vector<float> h(n16 * 4), r(n16 * 4);
    for (int i = 0; i < n16 * 4; ++i) h[i] = (float)i;
    float4 *din, *dout;
    CK(hipMalloc(&din, n16 * 16)); CK(hipMalloc(&dout, n16 * 16));
    CK(hipMemcpy(din, h.data(), n16 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, n16 * 16));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 16384, 0, din, dout, n16);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), dout, n16 * 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n16 * 4; ++i) {
        float want = h[i] + ((i & 3) == 0 ? 1.0f : 0.0f);
        if (r[i] != want) { if (bad < 5) printf("mismatch at %d: got %f want %f\n", i, r[i], want); ++bad; }
    }
    printf(bad ? "FAIL %d\n" : "glds ok\n", bad);
    return bad != 0;
}
