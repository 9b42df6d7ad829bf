// kernels_dpp.hip -- wave-autonomous row kernel for lengths N = R0 * 3 (BASELINE config 3: 93 = 31 * 3).
//
// The tile kernel runs 93-point rows as two passes with an LDS exchange between them: 64 rows x 744 B per workgroup,
// three workgroups (nine waves) per CU, two workgroup barriers and four LDS sweeps per tile (scatter, gather, scatter,
// flat copy for the coalesced store).  Here the radix-3 stage of the reference's [31, 3] plan (fft/fft/_fft.mojo:189-296
// with R = 3, P = 31) runs ACROSS THREE ADJACENT LANES through DPP row shifts -- the "wavefront shuffle for the small
// radices" of the north star -- and nothing is exchanged through memory between the two stages:
//
//   lane (row r, j in 0..2)   loads x[r][j + 3 m], m = 0..R0-1          (24-byte runs: measured nearly free for loads,
//                                                                        tools/micro/lane_chunks.hip)
//                             X_j = DFT_R0 of them in registers          (the radix-R0 stage, P = 1)
//                             Z_j[s] = X_j[s] * W_N^(j s)                (Stockham twiddle of the radix-3 stage)
//   the three lanes of a row  exchange Z by v_mov_b32_dpp row_shl / row_shr and EACH computes one output of the
//                             radix-3 butterfly:  Y[31 j + s] = Z_0[s] + W_3^j Z_1[s] + W_3^(2j) Z_2[s]
//   so lane j ends with the 31 CONSECUTIVE outputs 31 j .. 31 j + 30 of its row.
//
// Chunk-per-lane STORES are slow (4.1 TB/s against 5.6 coalesced, same microbenchmark), so the wave writes its rows
// in natural order into a wave-private LDS slab and streams that slab out as one linear, fully coalesced run.  A wave
// never waits for another wave: no workgroup barrier exists in the kernel, one LDS write + one LDS read per element
// instead of two + two, twelve independent one-wave workgroups per CU.
//
// MEASURED (500k x 93, MI355X): 0.163 ms against 0.143 ms for the two-pass tile kernel -- the shuffle stage costs more
// than the LDS exchange it removes.  Every lane computes ONE output of each radix-3 butterfly with three complex
// multiplies by lane-dependent constants (40 VALU operations per output against ~12 per output for a whole butterfly in
// one lane), the 31-point butterfly keeps the kernel at 156 VGPRs (three waves per SIMD, VALU ~40 % busy but no fourth
// wave to overlap with), and a quarter of the lanes idle (12 of 16 per DPP row; with 15 of 16 the wave's block is not a
// whole number of 128-byte lines and the store drops to 0.20 ms).  The kernel is therefore OPT-IN (MIFFT_DPP=1); it
// is parity-tested like every other path (tests/test_gpu_parity.py::test_dpp_radix3_rows).
//
// Lane layout: a DPP row has 16 lanes = RPW / 4 rows of the matrix x 3 lanes (+ idle lanes); a wave owns RPW rows.
// RPW = 16 (12 of 16 lanes busy) makes a wave's block 16 * 744 B = 93 whole 128-byte lines, so that its linear store
// never shares a line with another wave; RPW = 20 (15 of 16 lanes busy) does not.
#include "fast_table.h"
#include "mifft_config.h"

namespace mifft {

struct Dpp3Params {
    const void* in;
    void* out;
    const void* tw;  // plan table W_N^n (conjugated for inverse plans)
    long long n_rows;
    int inverse;
    float scale;
};

template <int CTRL>
MIFFT_DEV float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// value of the lane d places further up (row_shl) / down (row_shr) in the same 16-lane row
MIFFT_DEV cpx<float> from_lane_plus1(cpx<float> v) { return {dpp_row<0x101>(v.x), dpp_row<0x101>(v.y)}; }
MIFFT_DEV cpx<float> from_lane_plus2(cpx<float> v) { return {dpp_row<0x102>(v.x), dpp_row<0x102>(v.y)}; }
MIFFT_DEV cpx<float> from_lane_minus1(cpx<float> v) { return {dpp_row<0x111>(v.x), dpp_row<0x111>(v.y)}; }
MIFFT_DEV cpx<float> from_lane_minus2(cpx<float> v) { return {dpp_row<0x112>(v.x), dpp_row<0x112>(v.y)}; }

template <int R0, bool NT_STORE, int RPW>
__global__ __launch_bounds__(64, 3) void rows_x3_dpp_kernel(const Dpp3Params p) {
    using V = cpx<float>;
    constexpr int N = 3 * R0, RPR = RPW / 4;  // rows per DPP row
    static_assert(RPW % 4 == 0 && RPR * 3 <= 16, "RPW / 4 rows of three lanes per 16-lane DPP row");
    __shared__ __attribute__((aligned(16))) V slab[RPW * N];  // this wave's 20 rows, natural order
    __shared__ V ltw[3 * R0];                                  // W_N^(j s): row j = 0 is all ones
    const int lane = threadIdx.x;
    for (int e = lane; e < 3 * R0; e += 64) {
        const int j = e / R0, s = e - j * R0;
        V w = ((const V*)p.tw)[j * s];
        if (p.inverse) w.y = -w.y;  // the plan's table is conjugated for inverse plans; forward W here (conj trick below)
        ltw[e] = w;
    }
    const int sub = lane & 15, grp = sub / 3, j = sub - grp * 3;
    const bool active = sub < 3 * RPR;
    const int rl = (lane >> 4) * RPR + (active ? grp : 0);  // row inside the wave's block (idle lanes shadow a row)
    // this lane's radix-3 coefficients on (own, next position, position after next), positions counted mod 3:
    // output index = j, Y_j = sum_t W_3^(t j) Z_t with t = j, j + 1, j + 2
    const float h = -0.5f, q = -0.86602540378443864676f;  // W_3 = h + i q
    V c_own = {1.f, 0.f}, c_p1 = {1.f, 0.f}, c_p2 = {1.f, 0.f};
    if (j == 1) {  // t = 1, 2, 0 -> W^1, W^2, W^0
        c_own = {h, q};
        c_p1 = {h, -q};
    } else if (j == 2) {  // t = 2, 0, 1 -> W^4 = W^1, W^0, W^2
        c_own = {h, q};
        c_p2 = {h, -q};
    }
    __syncthreads();  // (one wave: orders the table fill; the only barrier of the kernel, outside the loop)

    const V* gin = (const V*)p.in;
    V* gout = (V*)p.out;
    const long long n_groups = (p.n_rows + RPW - 1) / RPW;
    for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        int lane_o = lane;  // opaque per group: keeps the address arithmetic out of the loop-invariant registers
        asm volatile("" : "+v"(lane_o));
        const long long row0 = g * RPW;
        long long row = row0 + rl;
        if (row >= p.n_rows) row = p.n_rows - 1;  // ragged last group: shadow a valid row, never stored
        const V* src = gin + row * N + j;
        V v[R0];
#pragma unroll
        for (int m = 0; m < R0; ++m) v[m] = src[3 * m];
        if (p.inverse) {
#pragma unroll
            for (int m = 0; m < R0; ++m) v[m].y = -v[m].y;
        }
        Dft<R0, float, 1>::run(v);
        V* dst = slab + rl * N + R0 * j;
#pragma unroll
        for (int s = 0; s < R0; ++s) {
            const V z = cmul(v[s], ltw[j * R0 + s]);
            const V a = from_lane_plus1(z), b = from_lane_plus2(z), c = from_lane_minus1(z), d = from_lane_minus2(z);
            // the lanes holding positions j + 1 and j + 2 (mod 3) of this row
            const V p1 = j == 2 ? d : a;
            const V p2 = j == 0 ? b : c;
            V y = cmul(z, c_own);
            const V t1 = cmul(p1, c_p1), t2 = cmul(p2, c_p2);
            y = y + t1;
            y = y + t2;
            if (active) dst[s] = y;
        }
        wave_lds_fence();
        const long long left = p.n_rows - row0;
        const int total = (int)(left < RPW ? left : RPW) * N;
        V* o = gout + row0 * N;
        const float sx = p.inverse ? p.scale : 1.0f, sy = p.inverse ? -p.scale : 1.0f;
        for (int i = lane_o; i < total; i += 64) {
            V y = slab[i];
            y.x *= sx;
            y.y *= sy;
            gstore<NT_STORE>(o + i, y);
        }
        wave_lds_fence();
    }
}

template <int R0, bool NT_STORE, int RPW>
static int launch_dpp3(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count, hipStream_t stream) {
    if (count == 0) return MIFFT_OK;
    Dpp3Params q{};
    q.in = in;
    q.out = out;
    q.tw = pass.d_twiddle;
    q.n_rows = count * pass.outer;
    q.inverse = plan.inverse;
    q.scale = plan.inverse ? (float)(1.0 / (double)pass.N) : 1.0f;
    const long long n_groups = (q.n_rows + RPW - 1) / RPW;
    long long per_cu = (160 * 1024) / ((long long)(RPW + 1) * 3 * R0 * 8);  // one-wave workgroups per CU by LDS ...
    if (per_cu > 12) per_cu = 12;                                             // ... and by registers (3 waves per SIMD)
    long long grid = (long long)plan.num_cus * per_cu;
    if (grid > n_groups) grid = n_groups;
    hipLaunchKernelGGL((rows_x3_dpp_kernel<R0, NT_STORE, RPW>), dim3((unsigned)grid), dim3(64), 0, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_error(e, "rows_x3_dpp_kernel launch");
    return MIFFT_OK;
}

// contiguous dimension of N = 3 * R0 points, complex fp32 in and out
bool select_dpp_rows(const Plan& plan, DimPass& pass) {
    if (!config().dpp) return false;  // opt-in (MIFFT_DPP=1, lab build): slower than the tile kernel (see the header)
    if (pass.inner != 1 || !pass.first || plan.out_dtype != MIFFT_F32 || plan.in_dtype != MIFFT_F32 ||
        plan.in_components != 2)
        return false;
    const bool streaming = plan.size_batch() * (double)plan.prod * (double)plan.out_elem_bytes() * 2.0 > config().streaming_min_bytes;
    // (16 rows per wave; the 20-row layout -- 15 of 16 lanes busy, blocks that are no whole number of lines -- measured
    //  0.20 ms against 0.163 and went with the MIFFT_DPP_VARIANT knob in round 3)
    const bool nt = streaming;
    const int rpw = 16;
    if (pass.N == 93) {
        pass.kernel_name = nt ? "rows93_31x3_dpp_nts" : "rows93_31x3_dpp";
        pass.launch = nt ? launch_dpp3<31, true, 16> : launch_dpp3<31, false, 16>;
    } else {
        return false;
    }
    pass.prepare = nullptr;
    pass.tile = rpw;
    pass.threads = 64;
    pass.lds_bytes = (size_t)(rpw * pass.N + pass.N) * 8;
    pass.ld = (int)pass.N;
    return true;
}

}  // namespace mifft
