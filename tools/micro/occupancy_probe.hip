// How many workgroups of a given LDS footprint does the runtime place on one CU?  (gfx950: 160 KiB per CU)
//   hipcc -O3 --offload-arch=gfx950 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float lds[];
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int n, long long* clk) {
    // records the start/stop clock of each workgroup and its CU id so that co-residency can be seen directly
    long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += THREADS) lds[i] = i;
    __syncthreads();
    float s = 0;
    for (int r = 0; r < 200; ++r)
        for (int i = threadIdx.x; i < n; i += THREADS) s += lds[(i * 7 + r) % n];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        clk[blockIdx.x * 3 + 0] = t0;
        clk[blockIdx.x * 3 + 1] = t1;
        clk[blockIdx.x * 3 + 2] = hwid;
    }
}
template <int THREADS>
void probe(size_t bytes) {
    int nb = -1;
    hipFuncSetAttribute((const void*)k<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<THREADS>, THREADS, bytes);
    float* out;
    long long* clk;
    const int grid = 512;
    hipMalloc(&out, (size_t)grid * THREADS * 4);
    hipMalloc(&clk, grid * 3 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<THREADS><<<grid, THREADS, bytes>>>(out, (int)(bytes / 4), clk);
    hipEventRecord(a);
    k<THREADS><<<grid, THREADS, bytes>>>(out, (int)(bytes / 4), clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static long long h[512 * 3];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    // count workgroups whose lifetime overlaps workgroup 0's on the same CU/SE id
    int overlap = 0;
    for (int i = 1; i < grid; ++i)
        if ((h[i * 3 + 2] & 0xffffff00) == (h[2] & 0xffffff00) && h[i * 3] < h[1] && h[i * 3 + 1] > h[0]) ++overlap;
    printf("threads %4d lds %7zu: occupancy API %d (%s), 512 WGs in %.3f ms, WGs overlapping WG0 on its CU: %d\n", THREADS,
           bytes, nb, hipGetErrorString(e), ms, overlap);
    hipFree(out);
    hipFree(clk);
}
int main() {
    for (size_t b : {163840ul, 81920ul, 81408ul, 80896ul, 79872ul, 65536ul, 54528ul, 40960ul}) probe<512>(b);
    for (size_t b : {81920ul, 40960ul, 32768ul}) probe<256>(b);
    return 0;
}
