/* cabi_demo.c -- drives libmifft.so through include/mifft.h from plain C: no Python, no torch.
 * Builds with hipcc (for hipMalloc / hipMemcpy only); checks a batched 1-D C2C fp32 transform and a 2-D
 * fp64 transform against a naive O(N^2) DFT in long double.  Exit code 0 = pass. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mifft.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define MIFFT_OK_(x) do { int r_ = (x); if (r_ < 0) { printf("mifft %d: %s\n", r_, mifft_last_error()); return 3; } } while (0)

static double check_1d_f32(void) {
    enum { B = 5, N = 1024 };
    static float x[B * N * 2], y[B * N * 2];
    unsigned s = 7;
    for (int i = 0; i < B * N * 2; ++i) { s = s * 1664525u + 1013904223u; x[i] = ((s >> 9) & 0xFFFF) / 65536.0f - 0.5f; }
    void *dx, *dy;
    if (hipMalloc(&dx, sizeof x) != hipSuccess || hipMalloc(&dy, sizeof y) != hipSuccess) return 1e9;
    hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice);
    mifft_plan* plan = NULL;
    int64_t dims[1] = {N};
    uint32_t bases[1] = {2};
    int32_t lens[1] = {1};
    if (mifft_plan_create(&plan, 0, MIFFT_F32, MIFFT_F32, 1, dims, B, 2, 0, bases, lens, MIFFT_FLAG_NONE) < 0) return 1e9;
    uint32_t stages[64];
    int ns = mifft_plan_stages(plan, 0, stages, 64);
    printf("1-D: kernel %s, %d user stages of radix %u, %d launch(es), %zu scratch bytes\n",
           mifft_plan_kernel_name(plan, 0), ns, stages[0], mifft_plan_num_launches(plan), mifft_plan_scratch_bytes(plan));
    if (mifft_plan_scratch_bytes(plan) != 0) { printf("a 1024-point plan must not own a scratch tensor\n"); return 1e9; }
    if (mifft_exec(plan, dx, dy, NULL) < 0) return 1e9;
    hipDeviceSynchronize();
    hipMemcpy(y, dy, sizeof y, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < B; b += 4)
        for (int k = 0; k < N; k += 37) {
            long double re = 0, im = 0, nrm = 0;
            for (int n = 0; n < N; ++n) {
                long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)((n * k) % N) / N, xr = x[(b * N + n) * 2], xi = x[(b * N + n) * 2 + 1];
                re += xr * cosl(a) - xi * sinl(a);
                im += xr * sinl(a) + xi * cosl(a);
                nrm += xr * xr + xi * xi;
            }
            double e = hypot((double)(re - y[(b * N + k) * 2]), (double)(im - y[(b * N + k) * 2 + 1])) / sqrt((double)nrm);
            if (e > worst) worst = e;
        }
    mifft_plan_destroy(plan);
    hipFree(dx);
    hipFree(dy);
    return worst;
}

static double check_2d_f64_inverse(void) {
    enum { B = 2, N0 = 12, N1 = 10 };
    static double x[B * N0 * N1 * 2], y[B * N0 * N1 * 2], z[B * N0 * N1 * 2];
    for (int i = 0; i < B * N0 * N1 * 2; ++i) x[i] = sin(0.37 * i) + 0.01 * i;
    void *dx, *dy, *dz;
    if (hipMalloc(&dx, sizeof x) != hipSuccess || hipMalloc(&dy, sizeof y) != hipSuccess || hipMalloc(&dz, sizeof z) != hipSuccess) return 1e9;
    hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice);
    int64_t dims[2] = {N0, N1};
    mifft_plan *fwd = NULL, *inv = NULL;
    if (mifft_plan_create(&fwd, 0, MIFFT_F64, MIFFT_F64, 2, dims, B, 2, 0, NULL, NULL, 0) < 0) return 1e9;
    if (mifft_plan_create(&inv, 0, MIFFT_F64, MIFFT_F64, 2, dims, B, 2, 1, NULL, NULL, 0) < 0) return 1e9;
    if (mifft_exec(fwd, dx, dy, NULL) < 0 || mifft_exec(inv, dy, dz, NULL) < 0) return 1e9;
    hipDeviceSynchronize();
    hipMemcpy(y, dy, sizeof y, hipMemcpyDeviceToHost);
    hipMemcpy(z, dz, sizeof z, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < B * N0 * N1 * 2; ++i) worst = fmax(worst, fabs(z[i] - x[i]));   /* round trip */
    /* DC bin of batch 1 = sum of its inputs */
    double sr = 0, si = 0;
    for (int i = 0; i < N0 * N1; ++i) { sr += x[(N0 * N1 + i) * 2]; si += x[(N0 * N1 + i) * 2 + 1]; }
    worst = fmax(worst, fmax(fabs(sr - y[N0 * N1 * 2]), fabs(si - y[N0 * N1 * 2 + 1])) / (fabs(sr) + fabs(si)));
    mifft_plan_destroy(fwd);
    mifft_plan_destroy(inv);
    hipFree(dx); hipFree(dy); hipFree(dz);
    return worst;
}

int main(void) {
    printf("libmifft version %d, %d HIP device(s)\n", mifft_version(), mifft_device_count());
    if (mifft_device_count() < 1) { printf("no device: libmifft has no CPU path\n"); return 4; }
    /* error reporting across the boundary */
    mifft_plan* bad = NULL;
    int64_t d32[1] = {32};
    uint32_t b42[2] = {4, 2};
    int32_t l2[1] = {2};
    int rc = mifft_plan_create(&bad, 0, MIFFT_F32, MIFFT_F32, 1, d32, 1, 2, 0, b42, l2, 0);
    printf("bases [4,2] for N=32 -> %d (%s): %s\n", rc, mifft_status_string(rc), mifft_last_error());
    if (rc != MIFFT_ERR_BAD_BASES || bad != NULL) return 5;
    double e1 = check_1d_f32(), e2 = check_2d_f64_inverse();
    printf("1-D fp32 rel err %.3e, 2-D fp64 round trip / DC err %.3e\n", e1, e2);
    return (e1 < 1e-5 && e2 < 1e-11) ? 0 : 1;
}
