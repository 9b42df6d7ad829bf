// vendor_fft_bench.cpp -- BENCH-ONLY comparator: the vendor FFT (rocFFT) on the shapes bench.py times, in its own
// process.  The analogue of the reference's cuFFT harness (cufft-benchmark-main/cufft_benchmark.cu:52-53,70-101: C2C
// fp32, out of place, "1 complex in + 1 complex out" bytes, planning time reported apart, 3 warm-up executions).
// It is never linked into libmifft.so and nothing under hackathon_fft_amd/ calls it.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -o vendor_fft_bench vendor_fft_bench.cpp -lrocfft
//   ./vendor_fft_bench [--iters K] 100000x1024 500000x93 100x640x480 10x128x128x128
// (first number = batch; one JSON line per shape on stdout)
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
#define CKR(x)                                                                         \
    do {                                                                               \
        rocfft_status s_ = (x);                                                        \
        if (s_ != rocfft_status_success) {                                             \
            fprintf(stderr, "rocFFT status %d at %s:%d\n", (int)s_, __FILE__, __LINE__); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static int run_shape(const std::string& spec, int iters) {
    std::vector<size_t> dims;  // batch, d0, d1, ... (row-major)
    {
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t nx = spec.find('x', pos);
            if (nx == std::string::npos) nx = spec.size();
            dims.push_back((size_t)atoll(spec.substr(pos, nx - pos).c_str()));
            pos = nx + 1;
        }
    }
    if (dims.size() < 2 || dims.size() > 4) {
        fprintf(stderr, "bad shape %s\n", spec.c_str());
        return 1;
    }
    const size_t batch = dims[0];
    size_t n = 1;
    for (size_t i = 1; i < dims.size(); ++i) n *= dims[i];
    const size_t bytes = batch * n * 2 * sizeof(float);
    void *x = nullptr, *y = nullptr;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    {  // N(0,1)-like synthetic input (values do not change the timing)
        std::vector<float> h(1 << 20);
        unsigned s = 1234;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 22);
        }
        for (size_t off = 0; off < bytes; off += h.size() * sizeof(float)) {
            const size_t c = std::min(h.size() * sizeof(float), bytes - off);
            CK(hipMemcpy((char*)x + off, h.data(), c, hipMemcpyHostToDevice));
        }
    }
    // rocFFT lengths are fastest-dimension first
    std::vector<size_t> lengths;
    for (size_t i = dims.size() - 1; i >= 1; --i) lengths.push_back(dims[i]);
    rocfft_plan plan = nullptr;
    const auto p0 = std::chrono::steady_clock::now();
    CKR(rocfft_plan_create(&plan, rocfft_placement_notinplace, rocfft_transform_type_complex_forward,
                           rocfft_precision_single, lengths.size(), lengths.data(), batch, nullptr));
    size_t work = 0;
    CKR(rocfft_plan_get_work_buffer_size(plan, &work));
    rocfft_execution_info info = nullptr;
    CKR(rocfft_execution_info_create(&info));
    void* wbuf = nullptr;
    if (work) {
        CK(hipMalloc(&wbuf, work));
        CKR(rocfft_execution_info_set_work_buffer(info, wbuf, work));
    }
    const double plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
    void* in[1] = {x};
    void* out[1] = {y};
    for (int i = 0; i < 3; ++i) CKR(rocfft_execute(plan, in, out, info));  // warm-up, as the reference harness
    CK(hipDeviceSynchronize());
    // the reference harness times ONE execution with a host clock; here: that, plus the average of `iters`
    // back-to-back executions between HIP events on the same (null) stream
    const auto t0 = std::chrono::steady_clock::now();
    CKR(rocfft_execute(plan, in, out, info));
    CK(hipDeviceSynchronize());
    const double one_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < iters; ++i) CKR(rocfft_execute(plan, in, out, info));  // clock ramp
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CKR(rocfft_execute(plan, in, out, info));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)iters;
    const double flops = 5.0 * (double)n * std::log2((double)n) * (double)batch;
    const double moved = 2.0 * (double)bytes;  // 1 complex in + 1 complex out (cufft_benchmark.cu:52-53)
    printf("{\"shape\": \"%s\", \"library\": \"rocFFT\", \"plan_ms\": %.2f, \"single_exec_host_clock_ms\": %.4f, "
           "\"ms\": %.5f, \"iters\": %d, \"gflops\": %.1f, \"gbs\": %.1f, \"work_buffer_bytes\": %zu}\n",
           spec.c_str(), plan_ms, one_ms, ms, iters, flops / (ms * 1e-3) / 1e9, moved / (ms * 1e-3) / 1e9, work);
    fflush(stdout);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)rocfft_execution_info_destroy(info);
    (void)rocfft_plan_destroy(plan);
    if (wbuf) (void)hipFree(wbuf);
    (void)hipFree(x);
    (void)hipFree(y);
    return 0;
}

int main(int argc, char** argv) {
    int iters = 50;
    std::vector<std::string> shapes;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc)
            iters = atoi(argv[++i]);
        else
            shapes.push_back(argv[i]);
    }
    if (shapes.empty()) shapes = {"500000x128", "100000x1024", "500000x93", "100x640x480", "10x128x128x128"};
    CKR(rocfft_setup());
    int rc = 0;
    for (const auto& s : shapes) rc |= run_shape(s, iters);
    (void)rocfft_cleanup();
    return rc;
}
