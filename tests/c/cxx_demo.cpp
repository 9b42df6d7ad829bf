// cxx_demo.cpp -- the C++ host mirror (include/mifft.hpp) from a plain g++ program: plan_fft / fft with the reference's
// names and error texts, a batched 1-D transform with user bases, a 2-D real-input transform with the default bases and a
// prime length (Rader), checked against a naive long-double DFT.  Exit code 0 = pass.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "mifft.hpp"

static bool expect_error(const char* what, const char* text, const std::vector<int64_t>& in, const std::vector<int64_t>& out,
                         const std::vector<std::vector<uint32_t>>* bases) {
    try {
        mifftxx::plan_fft(MIFFT_F32, MIFFT_F32, in, out, mifftxx::DeviceContext{}, bases);
    } catch (const mifftxx::Error& e) {
        const bool ok = std::string(e.what()).find(text) != std::string::npos;
        std::printf("%s -> \"%s\" %s\n", what, e.what(), ok ? "ok" : "UNEXPECTED TEXT");
        return ok;
    }
    std::printf("%s -> no error\n", what);
    return false;
}

static double run(const std::vector<int64_t>& layout_in, const std::vector<std::vector<uint32_t>>* bases, const char* expect_kernel) {
    std::vector<int64_t> layout_out = layout_in;
    layout_out.back() = 2;
    size_t n_in = 1, n_out = 1;
    for (int64_t d : layout_in) n_in *= (size_t)d;
    for (int64_t d : layout_out) n_out *= (size_t)d;
    std::vector<float> x(n_in), y(n_out);
    unsigned s = 11;
    for (float& v : x) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 9) & 0xFFFF) / 65536.0f - 0.5f;
    }
    void *dx = nullptr, *dy = nullptr;
    if (hipMalloc(&dx, n_in * 4) != hipSuccess || hipMalloc(&dy, n_out * 4) != hipSuccess) return 1e9;
    if (hipMemcpy(dx, x.data(), n_in * 4, hipMemcpyHostToDevice) != hipSuccess) return 1e9;
    mifftxx::DeviceContext ctx;
    mifftxx::Plan plan = mifftxx::plan_fft(MIFFT_F32, MIFFT_F32, layout_in, layout_out, ctx, bases);
    std::printf("kernel %s, %d launch(es)\n", plan.kernel_name((int)layout_in.size() - 3).c_str(), plan.num_launches());
    if (expect_kernel && plan.kernel_name((int)layout_in.size() - 3).find(expect_kernel) == std::string::npos) return 1e9;
    mifftxx::fft(dy, dx, ctx, plan);
    if (hipDeviceSynchronize() != hipSuccess) return 1e9;
    if (hipMemcpy(y.data(), dy, n_out * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1e9;
    (void)hipFree(dx);
    (void)hipFree(dy);
    // naive DFT over the LAST transformed dims of batch entry 0 (1-D or 2-D), a few output bins
    const int C = (int)layout_in.back();
    const int rank = (int)layout_in.size();
    const int64_t N2 = layout_in[rank - 2], N1 = rank == 4 ? layout_in[1] : 1;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    double worst = 0, norm = 0;
    for (int64_t k1 = 0; k1 < N1; k1 += (N1 > 3 ? 3 : 1))
        for (int64_t k2 = 0; k2 < N2; k2 += 7) {
            long double re = 0, im = 0;
            for (int64_t n1 = 0; n1 < N1; ++n1)
                for (int64_t n2 = 0; n2 < N2; ++n2) {
                    const long double a = -two_pi * ((long double)((n1 * k1) % N1) / N1 + (long double)((n2 * k2) % N2) / N2);
                    const long double xr = x[(n1 * N2 + n2) * C], xi = C == 2 ? x[(n1 * N2 + n2) * C + 1] : 0;
                    re += xr * cosl(a) - xi * sinl(a);
                    im += xr * sinl(a) + xi * cosl(a);
                }
            const double dr = y[(k1 * N2 + k2) * 2] - (double)re, di = y[(k1 * N2 + k2) * 2 + 1] - (double)im;
            worst = std::fmax(worst, dr * dr + di * di);
            norm = std::fmax(norm, (double)(re * re + im * im));
        }
    return std::sqrt(worst / norm);
}

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);  // a crash must not swallow the lines before it
    bool ok = true;
    // the reference's compile-time asserts as exceptions, before any device work (fft/fft/fft.mojo:22-46, _utils.mojo:206-220)
    ok &= expect_error("rank 2", "The rank should be bigger than 2.", {8, 2}, {8, 2}, nullptr);
    ok &= expect_error("C_in = 3", "The last dimension of in_layout should be 1 or 2", {4, 64, 3}, {4, 64, 2}, nullptr);
    ok &= expect_error("inner dimension of 1", "no inner dimension should be of size 1", {4, 1, 64, 2}, {4, 1, 64, 2}, nullptr);
    const std::vector<std::vector<uint32_t>> bad = {{3}};
    ok &= expect_error("bases {3} for 1024 points", "bases", {4, 1024, 2}, {4, 1024, 2}, &bad);
    const std::vector<std::vector<uint32_t>> two = {{2}};
    const double e1 = run({5, 1024, 2}, &two, "rows1024");          // batched 1-D, user bases [2] (radix-2 Stockham)
    const double e2 = run({3, 40, 48, 1}, nullptr, nullptr);        // 2-D real input, the reference's default bases
    const std::vector<std::vector<uint32_t>> p97 = {{97}};
    const double e3 = run({6, 97, 2}, &p97, "rader");               // a prime length: Rader's convolution in LDS
    std::printf("relative errors: %.2e %.2e %.2e\n", e1, e2, e3);
    ok &= e1 < 1e-5 && e2 < 1e-5 && e3 < 1e-5;
    const auto st = mifftxx::ordered_bases(480, {2, 3, 5});
    ok &= st.size() == 7 && st[0] == 5 && st[1] == 3;               // [5, 3, 2, 2, 2, 2, 2] like _build_ordered_bases
    std::printf(ok ? "cxx demo ok\n" : "cxx demo FAILED\n");
    return ok ? 0 : 1;
}
