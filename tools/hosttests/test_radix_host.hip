// Host-side check of the register butterflies (fft_radix.h) against a naive O(R^2) DFT in long double.
#include "../../hackathon_fft_amd/csrc/fft_radix.h"
#include <cstdio>
#include <cmath>
#include <cstdlib>
using namespace mifft;
template <int R, typename T> double check() {
    cpx<T> v[R]; long double xr[R], xi[R];
    srand(R * 7 + sizeof(T));
    for (int i = 0; i < R; ++i) { v[i].x = (T)(rand() / (double)RAND_MAX - 0.5); v[i].y = (T)(rand() / (double)RAND_MAX - 0.5); xr[i] = v[i].x; xi[i] = v[i].y; }
    Dft<R, T, 1>::run(v);
    double worst = 0, norm = 0;
    for (int k = 0; k < R; ++k) {
        long double sr = 0, si = 0;
        for (int n = 0; n < R; ++n) { long double a = -2.0L * M_PIl * (long double)((n * k) % R) / R; sr += xr[n] * cosl(a) - xi[n] * sinl(a); si += xr[n] * sinl(a) + xi[n] * cosl(a); }
        worst = fmax(worst, fmax(fabs((double)(sr - v[k].x)), fabs((double)(si - v[k].y)))); norm = fmax(norm, fmax(fabsl(sr), fabsl(si)));
    }
    return worst / norm;
}
template <int R> int one() {
    double ef = check<R, float>(), ed = check<R, double>();
    printf("R=%2d  f32 %.2e  f64 %.2e\n", R, ef, ed);
    return (ef < 3e-6 && ed < 1e-14) ? 0 : 1;
}
int main() {
    int bad = 0;
    bad += one<2>(); bad += one<3>(); bad += one<4>(); bad += one<5>(); bad += one<6>(); bad += one<7>(); bad += one<8>(); bad += one<9>();
    bad += one<10>(); bad += one<11>(); bad += one<12>(); bad += one<13>(); bad += one<15>(); bad += one<16>(); bad += one<20>(); bad += one<25>();
    bad += one<31>(); bad += one<32>(); bad += one<30>(); bad += one<17>(); bad += one<64>();
    printf(bad ? "FAIL\n" : "all ok\n");
    return bad;
}
