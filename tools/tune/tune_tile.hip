// tune_tile.hip -- times tile_kernel<> variants against each other on one device, in one
// process, interleaved (guide rule 24), and checks every variant's output against the first
// variant of its group.  Development tool only; not part of libmifft.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGROUP=<n> -o tune_tile tune_tile.hip && ./tune_tile
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../hackathon_fft_amd/csrc/tile_kernel.h"
#include "tile_kernel_experimental.h"

using namespace mifft;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Variant {
    std::string name;
    std::function<void(const void*, void*, const void*, long long /*batch*/, long long /*outer*/, long long /*inner*/)> run;
    size_t lds;
};

#if GROUP >= 30  // fp64 groups
using DT = double;
#else
using DT = float;
#endif
static int g_cus = 256;
static int g_wg_override = 0;
#ifdef MIFFT_STAMPS
static unsigned long long* g_stamps = nullptr;  // [grid][16] phase cycles of thread 0 (diagnostic build)
static long long g_last_grid = 0, g_last_tiles = 0;
#define STAMP_LDS 256
#else
#define STAMP_LDS 0
#endif

template <class C>
Variant make(const char* name) {
    Variant v;
    v.name = name;
    v.lds = C::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.store_lim = C::N / 2;  // (HS variants)
        tp.r2c_tw = tw;           // (R2C variants: timing only -- the table is the N-point one)
        tp.out_pitch = 2 * C::N;
        if (C::COLS) {
            tp.inner = inner;
            tp.tiles_per_outer = (inner + C::TILE - 1) / C::TILE;
            tp.n_tiles = batch * outer * tp.tiles_per_outer;
        } else {
            tp.n_rows = batch * outer;
            tp.inner = 1;
            tp.tiles_per_outer = 1;
            tp.n_tiles = (tp.n_rows + C::TILE - 1) / C::TILE;
        }
        auto k = tile_kernel<C>;
        static bool set = false;
        if (!set && C::LDS_BYTES > 64 * 1024) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
            set = true;
        }
        long long grid = tile_grid<C>(g_cus, tp.n_tiles, g_wg_override);
        static bool told = false;
        if (!told && getenv("TUNE_OCC")) {  // resident workgroups per CU as the runtime sees them
            int nb = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, C::THREADS, C::LDS_BYTES + STAMP_LDS));
            printf("occupancy: %d workgroups of %d threads, %zu B LDS per CU; grid %lld for %lld tiles\n", nb, C::THREADS,
                   (size_t)C::LDS_BYTES, grid, tp.n_tiles);
            told = true;
        }
#ifdef MIFFT_STAMPS
        tp.tcol = g_stamps;
        g_last_grid = grid;
        g_last_tiles = tp.n_tiles;
        static bool set2 = false;
        if (!set2) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES + STAMP_LDS));
            set2 = true;
        }
#endif
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::THREADS), C::LDS_BYTES + STAMP_LDS, 0, tp);
    };
    return v;
}

// ping-pong column tiles: two halves of C::THREADS threads share one LDS tile buffer (tile_kernel_pp)
template <class C>
Variant make_pp(const char* name) {
    Variant v;
    v.name = name;
    v.lds = C::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.inner = inner;
        tp.tiles_per_outer = (inner + C::TILE - 1) / C::TILE;
        tp.n_tiles = batch * outer * tp.tiles_per_outer;
        auto k = tile_kernel_pp<C>;
        static bool set = false;
        if (!set && C::LDS_BYTES > 64 * 1024) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
            set = true;
        }
        long long per_cu = (160 * 1024) / (long long)C::LDS_BYTES;
        if (per_cu > 1024 / C::THREADS) per_cu = 1024 / C::THREADS;
        if (per_cu < 1) per_cu = 1;
        if (g_wg_override > 0) per_cu = g_wg_override;
        long long grid = std::min<long long>((long long)g_cus * per_cu, tp.n_tiles);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(2 * C::THREADS), C::LDS_BYTES, 0, tp);
    };
    return v;
}
#define VP(NAME, ...) make_pp<TileCfg<__VA_ARGS__>>(NAME)

template <class C>
Variant make_dma(const char* name) {
    Variant v;
    v.name = name;
    v.lds = C::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.n_rows = batch * outer;
        tp.inner = 1;
        tp.tiles_per_outer = 1;
        tp.n_tiles = (tp.n_rows + C::TILE - 1) / C::TILE;
        auto k = tile_kernel_dma<C>;
        long long grid = tile_grid<C>(g_cus, tp.n_tiles, g_wg_override);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::THREADS), 0, 0, tp);  // static LDS
    };
    return v;
}

template <class C, int G>
Variant make_dma_split(const char* name) {
    Variant v;
    v.name = name;
    v.lds = C::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.n_rows = batch * outer;
        tp.inner = 1;
        tp.tiles_per_outer = 1;
        tp.n_tiles = (tp.n_rows + C::TILE - 1) / C::TILE;
        auto k = tile_kernel_dma_split<C, G>;
        long long grid = tile_grid<C>(g_cus, tp.n_tiles, g_wg_override);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::THREADS), 0, 0, tp);
    };
    return v;
}

template <class CR, class CC>
Variant make_plane(const char* name) {
    Variant v;
    v.name = name;
    v.lds = CR::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.store_lim = CC::N / 2;  // (HS column sides)
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.inner = CC::TILE;
        tp.tiles_per_outer = 1;
        tp.n_tiles = batch * outer;  // planes
        auto k = plane_kernel<CR, CC>;
        static bool set = false;
        if (!set && CR::LDS_BYTES > 64 * 1024) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR::LDS_BYTES));
            set = true;
        }
        long long grid = tile_grid<CR>(g_cus, tp.n_tiles, g_wg_override);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(CR::THREADS), CR::LDS_BYTES, 0, tp);
    };
    return v;
}

template <class CR, class CC, int PAD, bool WL = false>
Variant make_plane_wp(const char* name) {
    Variant v;
    v.name = name;
    using G = WavePlane<CR, CC, PAD>;
    v.lds = G::LDS_BYTES;
    v.run = [](const void* in, void* out, const void* tw, long long batch, long long outer, long long inner) {
        TileParams tp{};
        tp.store_lim = CC::N / 2;  // (HS column sides)
        tp.in = in;
        tp.out = out;
        tp.tw = tw;
        tp.inverse = 0;
        tp.scale = 1.0;
        tp.inner = CC::TILE;
        tp.tiles_per_outer = 1;
        tp.n_tiles = batch * outer;  // planes
        auto k = plane_kernel_wp<CR, CC, PAD, false, WL>;
        static bool set = false;
        if (!set && G::LDS_BYTES > 64 * 1024) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
            set = true;
        }
        long long per_cu = (160 * 1024) / (long long)G::LDS_BYTES;
        if (per_cu > 2048 / CR::THREADS) per_cu = 2048 / CR::THREADS;
        if (per_cu < 1) per_cu = 1;
        if (g_wg_override > 0) per_cu = g_wg_override;
        long long grid = std::min<long long>(g_cus * per_cu, tp.n_tiles);
#ifdef MIFFT_STAMPS
        tp.tcol = g_stamps;
        g_last_grid = grid;
        g_last_tiles = tp.n_tiles;
        static bool set2 = false;
        if (!set2) {
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES + STAMP_LDS));
            set2 = true;
        }
#endif
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(CR::THREADS), G::LDS_BYTES + STAMP_LDS, 0, tp);
    };
    return v;
}

#define V(NAME, ...) make<TileCfg<__VA_ARGS__>>(NAME)
// wave-owned sub-problems after pass 0 (TileCfg::WSUB)
#define VW(NAME, ...) make<TileCfg<__VA_ARGS__, 0, false, false, 0, false, float, true>>(NAME)
#define VWD(NAME, ...) make<TileCfg<__VA_ARGS__, 0, false, false, 0, false, double, true>>(NAME)
// explicit non-temporal mode: ... PF, then NT (0 none, 1 loads, 2 stores, 3 both)
#define VN(NAME, NT, ...) make<TileCfg<__VA_ARGS__, 0, false, false, NT>>(NAME)
// real input (C_in = 1) promoted in the pass-0 load: ... PF, then NT
#define VR(NAME, NT, ...) make<TileCfg<__VA_ARGS__, 0, true, false, NT>>(NAME)
// real input + half store (TileCfg::HS; store_lim = N / 2): the first pass in front of a Hermitian last pass
#define VRH(NAME, NT, T, ...) make<TileCfg<T, __VA_ARGS__, 0, true, false, NT, false, T, false, false, 0, false, true>>(NAME)
// packed real rows (TileCfg::R2C): N is HALF the row length; LAST_DIRECT must be false
#define VRP(NAME, NT, T, ...) make<TileCfg<T, __VA_ARGS__, 0, false, false, NT, false, T, false, false, 0, false, false, true>>(NAME)
// ... with an LDS row pad (ROWS: pitch = N + PAD)
#define VNP(NAME, NT, PAD, ...) make<TileCfg<__VA_ARGS__, PAD, false, false, NT>>(NAME)
// DMA-staged flat-copy rows: T N NP R0..R3 TILE THREADS TWMODE MINW
#define DS(NAME, G, NTM, T, N, NP, R0, R1, R2, R3, TILE, THR, TWM, MINW) \
    make_dma_split<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THR, false, false, false, TWM, MINW, false, 0, false, true, NTM>, G>(NAME)
#define D(NAME, NTM, T, N, NP, R0, R1, R2, R3, TILE, THR, TWM, MINW) \
    make_dma<TileCfg<T, N, NP, R0, R1, R2, R3, TILE, THR, false, false, false, TWM, MINW, false, 0, false, true, NTM>>(NAME)
#define PL(NAME, THR, MINW, PF, R0, R1, R2, R3, NP)                                                              \
    make_plane<TileCfg<float, 128, NP, R0, R1, R2, R3, 128, THR, false, true, false, TW_LDS, MINW, PF>,          \
               TileCfg<float, 128, NP, R0, R1, R2, R3, 128, THR, true, false, true, TW_LDS, MINW, false>>(NAME)

#define PLW(NAME, PN, PAD, THR, MINW, PF, R0, R1, R2, R3, NP)                                                    \
    make_plane_wp<TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, false, true, false, TW_LDS, MINW, PF>,         \
                  TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, true, false, true, TW_LDS, MINW, false>, PAD>(NAME)

// wave-private plane with the LAST column pass dealt over the whole workgroup (512-byte store runs, one more barrier)
#define PLWL(NAME, PN, PAD, THR, MINW, PF, R0, R1, R2, R3, NP)                                                   \
    make_plane_wp<TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, false, true, false, TW_LDS, MINW, PF>,         \
                  TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, true, false, true, TW_LDS, MINW, false>, PAD, true>(NAME)

// real input (rows side promotes) + half store (column side): the plane in front of a Hermitian last pass
#define HSCOL(PN, NP, R0, R1, R2, R3, THR, MINW) \
    TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, true, false, true, TW_LDS, MINW, false, 0, false, false, 0, false, float, false, false, 0, false, true>
#define PLNR(NAME, PN, THR, MINW, PF, R0, R1, R2, R3, NP)                                                        \
    make_plane<TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, false, true, false, TW_LDS, MINW, PF, 0, true>,   \
               HSCOL(PN, NP, R0, R1, R2, R3, THR, MINW)>(NAME)
#define PLWR(NAME, PN, PAD, THR, MINW, PF, R0, R1, R2, R3, NP)                                                   \
    make_plane_wp<TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, false, true, false, TW_LDS, MINW, PF, 0, true>, \
                  HSCOL(PN, NP, R0, R1, R2, R3, THR, MINW), PAD>(NAME)
#define PLN(NAME, PN, THR, MINW, PF, R0, R1, R2, R3, NP)                                                         \
    make_plane<TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, false, true, false, TW_LDS, MINW, PF>,            \
               TileCfg<float, PN, NP, R0, R1, R2, R3, PN, THR, true, false, true, TW_LDS, MINW, false>>(NAME)

int main(int argc, char** argv) {
    if (argc > 1) g_wg_override = atoi(argv[1]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;

#if GROUP == 1  // ---- config 2: 100k x 1024 rows ----
    const long long batch = getenv("TUNE_BATCH") ? atoll(getenv("TUNE_BATCH")) : 100000, outer = 1, inner = 1;
    const int N = 1024;
    std::vector<Variant> vs = {
        VN("4x4x8x8 t4 512 lds w2 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        // the same kernel forced under 80 / 64 VGPRs (82 as it stands: two workgroups per CU resident, four launched)
        VN("4x4x8x8 t4 512 lds w6 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 6, false),
        VN("4x4x8x8 t4 512 lds w8 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 8, false),
        VN("4x4x8x8 t2 256 lds w6 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 6, false),
        VN("4x4x8x8 t2 256 lds w4 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 4, false),
        VN("4x4x8x8 t4 512 lds w2 pf nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, true),
        VN("4x4x8x8 t4 512 lds w4 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 4, false),
        VN("4x4x8x8 t2 256 lds w4 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 4, false),
        VN("4x4x8x8 t8 1024 lds w4 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 8, 1024, false, true, true, TW_LDS, 4, false),
        VN("4x4x8x8 t4 512 reg w2 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_REG, 2, false),
        VN("8x4x4x8 t4 512 lds w2 nt3", 3, float, 1024, 4, 8, 4, 4, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("4x8x4x8 t4 512 lds w2 nt3", 3, float, 1024, 4, 4, 8, 4, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("8x8x4x4 t4 512 lds w2 nt3", 3, float, 1024, 4, 8, 8, 4, 4, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("4x4x4x16 t4 512 lds w2 nt3", 3, float, 1024, 4, 4, 4, 4, 16, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("8x8x16 t4 512 lds w2 nt3", 3, float, 1024, 3, 8, 8, 16, 1, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("4x16x16 t4 512 lds w2 nt3", 3, float, 1024, 3, 4, 16, 16, 1, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("16x8x8 t4 256 lds w4 nt3", 3, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VN("16x8x8 t4 256 reg w2 pf nt3", 3, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_REG, 2, true),
        VN("4x4x8x8 t4 512 lds w2 nt1", 1, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("4x4x8x8 t4 512 lds w2 nt0", 0, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("4x4x8x8 t4 512 lds w2 nt2", 2, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("16x8x8 t4 256 lds w4 nt0", 0, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VN("16x8x8 t4 256 lds w4 nt2", 2, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    };
#elif GROUP == 2  // ---- config 3: 500k x 93 (TUNE_BATCH=<rows> for other batch sizes) ----
    const long long batch = getenv("TUNE_BATCH") ? atoll(getenv("TUNE_BATCH")) : 500000, outer = 1, inner = 1;
    const int N = 93;
    std::vector<Variant> vs = {
        VN("31x3 t64 192 w3 nt2", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, false, false, TW_LDS, 3, false),
        VN("31x3 t64 192 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("31x3 t64 192 w3 nt3 fd", 3, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("31x3 t64 192 w3 nt0 fd", 0, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("31x3 t64 192 w2 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 2, false),
        VN("31x3 t64 192 w2 nt2 fd pf", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 2, true),
        VN("31x3 t32 96 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 3, false),
        VN("31x3 t128 384 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 128, 384, false, true, false, TW_LDS, 3, false),
        VN("31x3 t21 63 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 21, 63, false, true, false, TW_LDS, 3, false),
        VN("31x3 t42 126 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 42, 126, false, true, false, TW_LDS, 3, false),
        VN("31x3 t64 192 w3 nt2 fd reg", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_REG, 3, false),
        VN("31x3 t85 255 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 85, 255, false, true, false, TW_LDS, 3, false),
        // line-aligned tiles (16 rows = 93 lines of 128 B) that leave room for a fourth workgroup per CU
        VN("31x3 t48 144 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 48, 144, false, true, false, TW_LDS, 3, false),
        VN("31x3 t48 192 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 48, 192, false, true, false, TW_LDS, 3, false),
        VN("31x3 t32 128 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 32, 128, false, true, false, TW_LDS, 3, false),
        VN("31x3 t80 256 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 80, 256, false, true, false, TW_LDS, 3, false),
        VN("31x3 t64 256 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 64, 256, false, true, false, TW_LDS, 3, false),
    };
#elif GROUP == 3  // ---- 500k x 128 rows (config 1 shape, config 5 z axis; TUNE_BATCH=<rows>) ----
    const long long batch = getenv("TUNE_BATCH") ? atoll(getenv("TUNE_BATCH")) : 500000, outer = 1, inner = 1;
    const int N = 128;
    std::vector<Variant> vs = {
        V("8x4x4 t16 reg w1", float, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
        V("8x4x4 t16 reg w4 pf", float, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 4, true),
        V("8x16 t32 reg w2", float, 128, 2, 8, 16, 1, 1, 32, 256, false, true, true, TW_REG, 2, false),
        V("16x8 t32 reg w2", float, 128, 2, 16, 8, 1, 1, 32, 256, false, true, true, TW_REG, 2, false),
        V("8x16 t32 reg w2 ldsout", float, 128, 2, 8, 16, 1, 1, 32, 256, false, true, false, TW_REG, 2, false),
        V("16x8 t32 reg w2 ldsin", float, 128, 2, 16, 8, 1, 1, 32, 256, false, false, true, TW_REG, 2, false),
        VN("8x4x4 t16 reg w1 nt3", 3, float, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
        VN("8x4x4 t16 reg w1 nt2", 2, float, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
        VN("8x4x4 t16 reg w1 nt1", 1, float, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 1, false),
    };
#elif GROUP == 4  // ---- config 4 second pass: columns of 640, inner 480, 100 images ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        V("c640 4x4x8x5 t16 512 lds", float, 640, 4, 4, 4, 8, 5, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c640 4x4x8x5 t16 512 lds pf", float, 640, 4, 4, 4, 8, 5, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 4x4x8x5 t16 1024 lds", float, 640, 4, 4, 4, 8, 5, 16, 1024, true, true, true, TW_LDS, 1, false),
        V("c640 4x4x8x5 t16 1024 lds pf", float, 640, 4, 4, 4, 8, 5, 16, 1024, true, true, true, TW_LDS, 1, true),
        V("c640 4x4x8x5 t8 256 lds w2", float, 640, 4, 4, 4, 8, 5, 8, 256, true, true, true, TW_LDS, 2, false),
        V("c640 4x4x8x5 t8 256 lds w2 pf", float, 640, 4, 4, 4, 8, 5, 8, 256, true, true, true, TW_LDS, 2, true),
        V("c640 4x4x8x5 t8 512 lds w2", float, 640, 4, 4, 4, 8, 5, 8, 512, true, true, true, TW_LDS, 2, false),
        V("c640 4x4x8x5 t8 512 lds w2 pf", float, 640, 4, 4, 4, 8, 5, 8, 512, true, true, true, TW_LDS, 2, true),
        V("c640 10x8x8 t16 512 lds", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c640 10x8x8 t16 512 lds pf", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 8x8x10 t16 512 lds", float, 640, 3, 8, 8, 10, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c640 4x4x8x5 t16 768 lds", float, 640, 4, 4, 4, 8, 5, 16, 768, true, true, true, TW_LDS, 1, false),
        V("c640 4x4x8x5 t16 640 lds", float, 640, 4, 4, 4, 8, 5, 16, 640, true, true, true, TW_LDS, 1, false),
        V("c640 10x8x8 t16 512 glb w4", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_GLOBAL, 4, false),
        V("c640 8x8x10 t16 512 glb w4", float, 640, 3, 8, 8, 10, 1, 16, 512, true, true, true, TW_GLOBAL, 4, false),
        V("c640 4x4x8x5 t16 512 glb w4", float, 640, 4, 4, 4, 8, 5, 16, 512, true, true, true, TW_GLOBAL, 4, false),
        V("c640 5x8x4x4 t16 512 glb w4", float, 640, 4, 5, 8, 4, 4, 16, 512, true, true, true, TW_GLOBAL, 4, false),
        V("c640 10x8x8 t16 512 glb w2", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_GLOBAL, 2, false),
        V("c640 4x4x8x5 t16 1024 glb w4", float, 640, 4, 4, 4, 8, 5, 16, 1024, true, true, true, TW_GLOBAL, 4, false),
        V("c640 10x8x8 t16 1024 glb w4", float, 640, 3, 10, 8, 8, 1, 16, 1024, true, true, true, TW_GLOBAL, 4, false),
    };
#elif GROUP == 5  // ---- config 4 first pass: 64000 rows of 480 ----
    const long long batch = 64000, outer = 1, inner = 1;
    const int N = 480;
    std::vector<Variant> vs = {
        V("r480 10x6x8 t8 128 glb", float, 480, 3, 10, 6, 8, 1, 8, 128, false, true, true, TW_GLOBAL, 1, false),
        V("r480 10x6x8 t8 128 lds w2", float, 480, 3, 10, 6, 8, 1, 8, 128, false, true, true, TW_LDS, 2, false),
        V("r480 10x6x8 t8 256 lds w2", float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        V("r480 10x6x8 t16 256 lds w2", float, 480, 3, 10, 6, 8, 1, 16, 256, false, true, true, TW_LDS, 2, false),
        V("r480 10x6x8 t4 128 lds w3", float, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 3, false),
        V("r480 8x6x10 t8 256 lds w2", float, 480, 3, 8, 6, 10, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        V("r480 16x30 t8 256 lds w2", float, 480, 2, 16, 30, 1, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        V("r480 4x4x30 t8 256 lds w2", float, 480, 3, 4, 4, 30, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        V("r480 10x6x8 t8 256 lds w2 pf", float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        V("r480 8x6x10 t8 256 lds w2 pf", float, 480, 3, 8, 6, 10, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VN("r480 10x6x8 t8 256 pf nt1", 1, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VN("r480 4x4x5x6 t8 512 w2 nt1", 1, float, 480, 4, 4, 4, 5, 6, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 5x4x4x6 t8 512 w2 nt1", 1, float, 480, 4, 5, 4, 4, 6, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 6x5x4x4 t8 512 w2 nt1", 1, float, 480, 4, 6, 5, 4, 4, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 4x5x4x6 t8 512 w2 nt1", 1, float, 480, 4, 4, 5, 4, 6, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 4x4x5x6 t16 1024 w4 nt1", 1, float, 480, 4, 4, 4, 5, 6, 16, 1024, false, true, true, TW_LDS, 4, false),
        VN("r480 4x4x5x6 t4 256 w2 nt1", 1, float, 480, 4, 4, 4, 5, 6, 4, 256, false, true, true, TW_LDS, 2, false),
        VN("r480 8x6x10 t8 512 w2 nt1", 1, float, 480, 3, 8, 6, 10, 1, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 10x6x8 t8 384 w2 nt1", 1, float, 480, 3, 10, 6, 8, 1, 8, 384, false, true, true, TW_LDS, 2, false),
        VN("r480 10x6x8 t16 512 w2 nt1", 1, float, 480, 3, 10, 6, 8, 1, 16, 512, false, true, true, TW_LDS, 2, false),
        VN("r480 6x8x10 t8 512 w2 nt1", 1, float, 480, 3, 6, 8, 10, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 27  // ---- real input, half store: 64000 rows of 480 (first pass of 100 x 640 x 480 real) ----
    const long long batch = 64000, outer = 1, inner = 1;
    const int N = 480;
    std::vector<Variant> vs = {
        VR("real full 10x6x8 t8 256 pf nt1", 1, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VRH("hs 10x6x8 t8 256 pf nt1 (shipped)", 1, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VRH("hs 10x6x8 t8 256 pf nt0", 0, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VRH("hs 10x6x8 t8 256 nt1", 1, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        VRH("hs 10x6x8 t4 128 pf nt1", 1, float, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 2, true),
        VRH("hs 10x6x8 t4 128 nt1", 1, float, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 2, false),
        VRH("hs 10x6x8 t16 512 nt1", 1, float, 480, 3, 10, 6, 8, 1, 16, 512, false, true, true, TW_LDS, 2, false),
        VRH("hs 8x6x10 t8 256 pf nt1", 1, float, 480, 3, 8, 6, 10, 1, 8, 256, false, true, true, TW_LDS, 2, true),
        VRH("hs 4x4x5x6 t8 512 nt1", 1, float, 480, 4, 4, 4, 5, 6, 8, 512, false, true, true, TW_LDS, 2, false),
        VRH("hs 6x5x4x4 t8 512 nt1", 1, float, 480, 4, 6, 5, 4, 4, 8, 512, false, true, true, TW_LDS, 2, false),
        VRH("hs 16x30 t8 256 nt1", 1, float, 480, 2, 16, 30, 1, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        VRH("hs 10x6x8 t8 256 pf nt1 flat-store", 1, float, 480, 3, 10, 6, 8, 1, 8, 256, false, true, false, TW_LDS, 2, true),
        VRP("r2c 15x8x2 t8 256 nt1", 1, float, 240, 3, 15, 8, 2, 1, 8, 256, false, true, false, TW_LDS, 2, false),
        VRP("r2c 15x8x2 t8 256 pf nt1", 1, float, 240, 3, 15, 8, 2, 1, 8, 256, false, true, false, TW_LDS, 2, true),
        VRP("r2c 16x15 t4 128 nt1", 1, float, 240, 2, 16, 15, 1, 1, 4, 128, false, true, false, TW_LDS, 2, false),
        VRP("r2c 8x6x5 t16 256 nt1", 1, float, 240, 3, 8, 6, 5, 1, 16, 256, false, true, false, TW_LDS, 2, false),
    };
#elif GROUP == 6  // ---- config 5 y / x axes: columns of 128 ----
    const long long batch = 10, outer = 128, inner = 128;  // y axis; x axis is outer 1, inner 16384
    const int N = 128;
    std::vector<Variant> vs = {
        V("c128 16x8 t16 128 reg", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_REG, 1, false),
        V("c128 16x8 t16 128 reg w3", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_REG, 3, false),
        V("c128 16x8 t16 128 lds w4", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_LDS, 4, false),
        V("c128 8x16 t16 128 lds w4", float, 128, 2, 8, 16, 1, 1, 16, 128, true, true, true, TW_LDS, 4, false),
        V("c128 8x4x4 t16 256 lds w4", float, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t32 256 lds w4", float, 128, 2, 16, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t8 64 lds w4", float, 128, 2, 16, 8, 1, 1, 8, 64, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t16 128 lds w4 pf", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_LDS, 4, true),
    };
#elif GROUP == 8  // ---- four-step passes: columns of 1024 over 1024 columns, 64 transforms of 2^20 ----
    const long long batch = 64, outer = 1, inner = 1024;
    const int N = 1024;
    std::vector<Variant> vs = {
        V("c1024 4x4x8x8 t8 512 lds", float, 1024, 4, 4, 4, 8, 8, 8, 512, true, true, true, TW_LDS, 1, false),
        V("c1024 4x4x8x8 t8 512 lds pf", float, 1024, 4, 4, 4, 8, 8, 8, 512, true, true, true, TW_LDS, 1, true),
        V("c1024 4x4x8x8 t16 1024 lds", float, 1024, 4, 4, 4, 8, 8, 16, 1024, true, true, true, TW_LDS, 1, false),
        V("c1024 4x4x8x8 t16 512 lds", float, 1024, 4, 4, 4, 8, 8, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c1024 16x8x8 t16 1024 lds", float, 1024, 3, 16, 8, 8, 1, 16, 1024, true, true, true, TW_LDS, 1, false),
        V("c1024 16x8x8 t16 512 lds", float, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c1024 8x8x16 t16 1024 lds", float, 1024, 3, 8, 8, 16, 1, 16, 1024, true, true, true, TW_LDS, 1, false),
        V("c1024 16x8x8 t8 512 lds", float, 1024, 3, 16, 8, 8, 1, 8, 512, true, true, true, TW_LDS, 1, false),
        V("c1024 16x8x8 t8 512 lds pf", float, 1024, 3, 16, 8, 8, 1, 8, 512, true, true, true, TW_LDS, 1, true),
        V("c1024 16x8x8 t8 256 lds w2", float, 1024, 3, 16, 8, 8, 1, 8, 256, true, true, true, TW_LDS, 2, false),
        V("c1024 4x4x8x8 t8 256 lds w2", float, 1024, 4, 4, 4, 8, 8, 8, 256, true, true, true, TW_LDS, 2, false),
        V("c1024 4x4x8x8 t4 256 lds w2", float, 1024, 4, 4, 4, 8, 8, 4, 256, true, true, true, TW_LDS, 2, false),
        V("c1024 16x8x8 t16 1024 lds pf", float, 1024, 3, 16, 8, 8, 1, 16, 1024, true, true, true, TW_LDS, 1, true),
        V("c1024 4x4x8x8 t16 1024 lds pf", float, 1024, 4, 4, 4, 8, 8, 16, 1024, true, true, true, TW_LDS, 1, true),
        V("c1024 8x8x16 t16 1024 lds pf", float, 1024, 3, 8, 8, 16, 1, 16, 1024, true, true, true, TW_LDS, 1, true),
        V("c1024 16x8x8 t16 512 lds pf", float, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c1024 16x8x8 t16 1024 glob pf", float, 1024, 3, 16, 8, 8, 1, 16, 1024, true, true, true, TW_GLOBAL, 1, true),
    };
#elif GROUP == 9  // ---- rows of 256, 400k transforms ----
    const long long batch = 400000, outer = 1, inner = 1;
    const int N = 256;
    std::vector<Variant> vs = {
        VN("16x16 t16 256 lds w2 nt0", 0, float, 256, 2, 16, 16, 1, 1, 16, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16 t16 256 lds w2 nt3", 3, float, 256, 2, 16, 16, 1, 1, 16, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16 t16 256 reg w2 nt3", 3, float, 256, 2, 16, 16, 1, 1, 16, 256, false, true, true, TW_REG, 2, false),
        VN("4x8x8 t16 512 lds w2 nt3", 3, float, 256, 3, 4, 8, 8, 1, 16, 512, false, true, true, TW_LDS, 2, false),
        VN("4x8x8 t16 256 lds w4 nt3", 3, float, 256, 3, 4, 8, 8, 1, 16, 256, false, true, true, TW_LDS, 4, false),
        VN("8x8x4 t16 512 lds w2 nt3", 3, float, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
        VN("4x4x4x4 t16 512 lds w2 nt3", 3, float, 256, 4, 4, 4, 4, 4, 16, 512, false, true, true, TW_LDS, 2, false),
        VN("4x4x4x4 t16 1024 lds w4 nt3", 3, float, 256, 4, 4, 4, 4, 4, 16, 1024, false, true, true, TW_LDS, 4, false),
        VN("8x8x4 t8 256 lds w4 nt3", 3, float, 256, 3, 8, 8, 4, 1, 8, 256, false, true, true, TW_LDS, 4, false),
        VN("8x8x4 t16 512 lds w2 nt0", 0, float, 256, 3, 8, 8, 4, 1, 16, 512, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 10  // ---- rows of 512, 200k transforms ----
    const long long batch = 200000, outer = 1, inner = 1;
    const int N = 512;
    std::vector<Variant> vs = {
        VN("8x8x8 t4 256 lds w4 nt0", 0, float, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VN("8x8x8 t4 256 lds w4 nt3", 3, float, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VN("8x8x8 t8 512 lds w2 nt3", 3, float, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("8x8x8 t8 256 lds w2 nt3", 3, float, 512, 3, 8, 8, 8, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        VN("4x4x4x8 t8 512 lds w2 nt3", 3, float, 512, 4, 4, 4, 4, 8, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("16x8x4 t8 256 lds w2 nt3", 3, float, 512, 3, 16, 8, 4, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        VN("16x32 t8 256 lds w2 nt3", 3, float, 512, 2, 16, 32, 1, 1, 8, 256, false, true, true, TW_LDS, 2, false),
        VN("8x8x8 t4 256 reg w2 pf nt3", 3, float, 512, 3, 8, 8, 8, 1, 4, 256, false, true, true, TW_REG, 2, true),
        VN("8x8x8 t8 512 lds w2 nt0", 0, float, 512, 3, 8, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 11  // ---- rows of 2048, 50k transforms ----
    const long long batch = 50000, outer = 1, inner = 1;
    const int N = 2048;
    std::vector<Variant> vs = {
        VN("16x16x8 t2 256 lds w2 nt0", 0, float, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16x8 t2 256 lds w2 nt3", 3, float, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16x8 t2 256 lds w4 nt3", 3, float, 2048, 3, 16, 16, 8, 1, 2, 256, false, true, true, TW_LDS, 4, false),
        VN("4x8x8x8 t2 512 lds w2 nt3", 3, float, 2048, 4, 4, 8, 8, 8, 2, 512, false, true, true, TW_LDS, 2, false),
        VN("4x8x8x8 t4 1024 lds w4 nt3", 3, float, 2048, 4, 4, 8, 8, 8, 4, 1024, false, true, true, TW_LDS, 4, false),
        VN("4x8x8x8 t1 256 lds w4 nt3", 3, float, 2048, 4, 4, 8, 8, 8, 1, 256, false, true, true, TW_LDS, 4, false),
        VN("16x16x8 t1 128 lds w4 nt3", 3, float, 2048, 3, 16, 16, 8, 1, 1, 128, false, true, true, TW_LDS, 4, false),
        VN("8x16x16 t2 256 lds w4 nt3", 3, float, 2048, 3, 8, 16, 16, 1, 2, 256, false, true, true, TW_LDS, 4, false),
        VN("4x8x8x8 t2 512 lds w2 nt0", 0, float, 2048, 4, 4, 8, 8, 8, 2, 512, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 12  // ---- rows of 4096, 25k transforms ----
    const long long batch = 25000, outer = 1, inner = 1;
    const int N = 4096;
    std::vector<Variant> vs = {
        VN("16x16x16 t1 256 lds w2 nt0", 0, float, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16x16 t1 256 lds w2 nt3", 3, float, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 2, false),
        VN("16x16x16 t1 256 lds w4 nt3", 3, float, 4096, 3, 16, 16, 16, 1, 1, 256, false, true, true, TW_LDS, 4, false),
        VN("8x8x8x8 t1 512 lds w2 nt3", 3, float, 4096, 4, 8, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 2, false),
        VN("8x8x8x8 t2 1024 lds w4 nt3", 3, float, 4096, 4, 8, 8, 8, 8, 2, 1024, false, true, true, TW_LDS, 4, false),
        VN("8x8x8x8 t1 256 lds w4 nt3", 3, float, 4096, 4, 8, 8, 8, 8, 1, 256, false, true, true, TW_LDS, 4, false),
        VN("4x4x16x16 t1 512 lds w2 nt3", 3, float, 4096, 4, 4, 4, 16, 16, 1, 512, false, true, true, TW_LDS, 2, false),
        VN("16x16x16 t2 512 lds w2 nt3", 3, float, 4096, 3, 16, 16, 16, 1, 2, 512, false, true, true, TW_LDS, 2, false),
        VN("8x8x8x8 t1 512 lds w2 nt0", 0, float, 4096, 4, 8, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 7  // ---- config 5 fused z+y plane pass: 1280 planes of 128x128 ----
    const long long batch = 10, outer = 128, inner = 1;
    const int N = 128;  // tensor = batch*outer planes of 128x128 -> elems = batch*outer*128*128
    std::vector<Variant> vs = {
        PL("plane 8x4x4 1024 w4", 1024, 4, false, 8, 4, 4, 1, 3),
        PL("plane 8x4x4 1024 w4 pf", 1024, 4, true, 8, 4, 4, 1, 3),
        PL("plane 8x4x4 512 w2", 512, 2, false, 8, 4, 4, 1, 3),
        PL("plane 8x4x4 512 w2 pf", 512, 2, true, 8, 4, 4, 1, 3),
        PL("plane 16x8 512 w2", 512, 2, false, 16, 8, 1, 1, 2),
        PL("plane 16x8 512 w2 pf", 512, 2, true, 16, 8, 1, 1, 2),
        PL("plane 16x8 1024 w4", 1024, 4, false, 16, 8, 1, 1, 2),
        PL("plane 8x16 512 w2 pf", 512, 2, true, 8, 16, 1, 1, 2),
        PL("plane 4x4x8 1024 w4", 1024, 4, false, 4, 4, 8, 1, 3),
        PL("plane 4x4x8 512 w2 pf", 512, 2, true, 4, 4, 8, 1, 3),
        PL("plane 8x4x4 256 w1 pf", 256, 1, true, 8, 4, 4, 1, 3),
        PL("plane 16x8 1024 w4 pf", 1024, 4, true, 16, 8, 1, 1, 2),
        PL("plane 8x16 1024 w4", 1024, 4, false, 8, 16, 1, 1, 2),
        PL("plane 8x16 1024 w4 pf", 1024, 4, true, 8, 16, 1, 1, 2),
        PLW("wp 16x8 1024 w4 pf pad8", 128, 8, 1024, 4, true, 16, 8, 1, 1, 2),
        PLW("wp 16x8 1024 w4 pad8", 128, 8, 1024, 4, false, 16, 8, 1, 1, 2),
        PLW("wp 16x8 1024 w4 pf pad0", 128, 0, 1024, 4, true, 16, 8, 1, 1, 2),
        PLW("wp 8x16 1024 w4 pf pad8", 128, 8, 1024, 4, true, 8, 16, 1, 1, 2),
        PLW("wp 8x4x4 1024 w4 pf pad8", 128, 8, 1024, 4, true, 8, 4, 4, 1, 3),
        PLW("wp 4x4x8 1024 w4 pf pad8", 128, 8, 1024, 4, true, 4, 4, 8, 1, 3),
        PLW("wp 16x8 512 w2 pf pad8", 128, 8, 512, 2, true, 16, 8, 1, 1, 2),
        PLW("wp 16x8 512 w2 pf pad16", 128, 16, 512, 2, true, 16, 8, 1, 1, 2),
        PLW("wp 8x16 512 w2 pf pad8", 128, 8, 512, 2, true, 8, 16, 1, 1, 2),
        PLWL("wpwl 8x16 1024 w4 pf pad8", 128, 8, 1024, 4, true, 8, 16, 1, 1, 2),
        PLWL("wpwl 16x8 1024 w4 pf pad8", 128, 8, 1024, 4, true, 16, 8, 1, 1, 2),
        PLWL("wpwl 8x16 1024 w4 pf pad0", 128, 0, 1024, 4, true, 8, 16, 1, 1, 2),
        PLWL("wpwl 8x16 512 w2 pf pad8", 128, 8, 512, 2, true, 8, 16, 1, 1, 2),
        PLWL("wpwl 8x4x4 1024 w4 pf pad8", 128, 8, 1024, 4, true, 8, 4, 4, 1, 3),
    };
#elif GROUP == 13  // ---- 100 x 64^3: fused y+x planes, 6400 planes of 64x64 ----
    const long long batch = 100, outer = 64, inner = 1;
    const int N = 64;
    std::vector<Variant> vs = {
        PLN("plane64 4x4x4 512 w2 pf", 64, 512, 2, true, 4, 4, 4, 1, 3),
        PLN("plane64 4x4x4 512 w2", 64, 512, 2, false, 4, 4, 4, 1, 3),
        PLN("plane64 4x4x4 256 w2 pf", 64, 256, 2, true, 4, 4, 4, 1, 3),
        PLN("plane64 4x4x4 256 w4", 64, 256, 4, false, 4, 4, 4, 1, 3),
        PLN("plane64 8x8 512 w2", 64, 512, 2, false, 8, 8, 1, 1, 2),
        PLN("plane64 8x8 512 w2 pf", 64, 512, 2, true, 8, 8, 1, 1, 2),
        PLN("plane64 8x8 256 w2", 64, 256, 2, false, 8, 8, 1, 1, 2),
        PLN("plane64 8x8 256 w2 pf", 64, 256, 2, true, 8, 8, 1, 1, 2),
        PLN("plane64 8x8 256 w4", 64, 256, 4, false, 8, 8, 1, 1, 2),
        PLN("plane64 16x4 256 w2", 64, 256, 2, false, 16, 4, 1, 1, 2),
        PLN("plane64 4x16 256 w2", 64, 256, 2, false, 4, 16, 1, 1, 2),
        PLN("plane64 8x8 1024 w4", 64, 1024, 4, false, 8, 8, 1, 1, 2),
        PLW("wp64 8x8 512 w2 pad8", 64, 8, 512, 2, false, 8, 8, 1, 1, 2),
        PLW("wp64 8x8 512 w2 pf pad8", 64, 8, 512, 2, true, 8, 8, 1, 1, 2),
        PLW("wp64 8x8 256 w2 pf pad16", 64, 16, 256, 2, true, 8, 8, 1, 1, 2),
        PLW("wp64 8x8 256 w4 pad16", 64, 16, 256, 4, false, 8, 8, 1, 1, 2),
        PLW("wp64 4x4x4 512 w2 pf pad8", 64, 8, 512, 2, true, 4, 4, 4, 1, 3),
    };
#elif GROUP == 28  // ---- 100 x 64^3 REAL input: fused y+x planes with the half store, 6400 planes of 64x64 ----
    const long long batch = 100, outer = 64, inner = 1;
    const int N = 64;
    std::vector<Variant> vs = {
        PLNR("plane64 r hs 8x8 512 w2 (shipped)", 64, 512, 2, false, 8, 8, 1, 1, 2),
        PLNR("plane64 r hs 8x8 512 w2 pf", 64, 512, 2, true, 8, 8, 1, 1, 2),
        PLNR("plane64 r hs 8x8 256 w2", 64, 256, 2, false, 8, 8, 1, 1, 2),
        PLNR("plane64 r hs 8x8 256 w4", 64, 256, 4, false, 8, 8, 1, 1, 2),
        PLNR("plane64 r hs 4x4x4 512 w2", 64, 512, 2, false, 4, 4, 4, 1, 3),
        PLNR("plane64 r hs 16x4 256 w2", 64, 256, 2, false, 16, 4, 1, 1, 2),
        PLNR("plane64 r hs 8x8 1024 w4", 64, 1024, 4, false, 8, 8, 1, 1, 2),
        PLWR("wp64 r hs 8x8 512 w2 pad8", 64, 8, 512, 2, false, 8, 8, 1, 1, 2),
        PLWR("wp64 r hs 8x8 512 w2 pf pad8", 64, 8, 512, 2, true, 8, 8, 1, 1, 2),
        PLWR("wp64 r hs 8x8 256 w2 pf pad16", 64, 16, 256, 2, true, 8, 8, 1, 1, 2),
        PLWR("wp64 r hs 8x8 256 w4 pad16", 64, 16, 256, 4, false, 8, 8, 1, 1, 2),
        PLWR("wp64 r hs 4x4x4 512 w2 pf pad8", 64, 8, 512, 2, true, 4, 4, 4, 1, 3),
    };
#elif GROUP == 29  // ---- 10 x 128^3 REAL input: fused planes with the half store, 1280 planes of 128x128 ----
    const long long batch = 10, outer = 128, inner = 1;
    const int N = 128;
    std::vector<Variant> vs = {
        PLWR("wp r hs 8x16 1024 w4 pf pad8 (shipped)", 128, 8, 1024, 4, true, 8, 16, 1, 1, 2),
        PLWR("wp r hs 8x16 1024 w4 pad8", 128, 8, 1024, 4, false, 8, 16, 1, 1, 2),
        PLWR("wp r hs 16x8 1024 w4 pf pad8", 128, 8, 1024, 4, true, 16, 8, 1, 1, 2),
        PLWR("wp r hs 8x4x4 1024 w4 pf pad8", 128, 8, 1024, 4, true, 8, 4, 4, 1, 3),
        PLWR("wp r hs 4x4x8 1024 w4 pf pad8", 128, 8, 1024, 4, true, 4, 4, 8, 1, 3),
        PLWR("wp r hs 8x16 512 w2 pf pad8", 128, 8, 512, 2, true, 8, 16, 1, 1, 2),
        PLWR("wp r hs 16x8 512 w2 pf pad8", 128, 8, 512, 2, true, 16, 8, 1, 1, 2),
        PLWR("wp r hs 8x16 1024 w4 pf pad0", 128, 0, 1024, 4, true, 8, 16, 1, 1, 2),
        PLNR("plane r hs 8x16 1024 w4 pf", 128, 1024, 4, true, 8, 16, 1, 1, 2),
        PLNR("plane r hs 16x8 1024 w4 pf", 128, 1024, 4, true, 16, 8, 1, 1, 2),
    };
#elif GROUP == 14  // ---- long rows: 3906 x 8192 ----
    const long long batch = 3906, outer = 1, inner = 1;
    const int N = 8192;
    std::vector<Variant> vs = {
        V("r8192 16x8x8x8 512 lds", float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 1, false),
        V("r8192 16x8x8x8 512 lds pf", float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 1, true),
        V("r8192 16x8x8x8 512 lds w2 pf", float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 2, true),
        V("r8192 8x8x8x16 512 lds pf", float, 8192, 4, 8, 8, 8, 16, 1, 512, false, true, true, TW_LDS, 1, true),
        V("r8192 16x8x8x8 1024 lds", float, 8192, 4, 16, 8, 8, 8, 1, 1024, false, true, true, TW_LDS, 1, false),
        V("r8192 16x8x8x8 1024 lds pf", float, 8192, 4, 16, 8, 8, 8, 1, 1024, false, true, true, TW_LDS, 1, true),
        V("r8192 16x8x8x8 256 lds w2", float, 8192, 4, 16, 8, 8, 8, 1, 256, false, true, true, TW_LDS, 2, false),
        V("r8192 16x8x8x8 512 glb w2", float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_GLOBAL, 2, false),
        V("r8192 16x8x8x8 512 glb w2 pf", float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_GLOBAL, 2, true),
        VN("r8192 16x8x8x8 512 lds pf nt3", 3, float, 8192, 4, 16, 8, 8, 8, 1, 512, false, true, true, TW_LDS, 1, true),
    };
#elif GROUP == 15  // ---- long rows: 1953 x 16384 ----
    const long long batch = 1953, outer = 1, inner = 1;
    const int N = 16384;
    std::vector<Variant> vs = {
        V("r16384 16x16x8x8 1024 glb w4", float, 16384, 4, 16, 16, 8, 8, 1, 1024, false, true, true, TW_GLOBAL, 4, false),
        V("r16384 16x16x8x8 1024 glb w4 pf", float, 16384, 4, 16, 16, 8, 8, 1, 1024, false, true, true, TW_GLOBAL, 4, true),
        V("r16384 8x8x16x16 1024 glb w4 pf", float, 16384, 4, 8, 8, 16, 16, 1, 1024, false, true, true, TW_GLOBAL, 4, true),
        V("r16384 16x16x8x8 512 glb w2", float, 16384, 4, 16, 16, 8, 8, 1, 512, false, true, true, TW_GLOBAL, 2, false),
        V("r16384 16x16x8x8 512 glb w2 pf", float, 16384, 4, 16, 16, 8, 8, 1, 512, false, true, true, TW_GLOBAL, 2, true),
        V("r16384 16x16x16x4 1024 glb w4 pf", float, 16384, 4, 16, 16, 16, 4, 1, 1024, false, true, true, TW_GLOBAL, 4, true),
        VN("r16384 16x16x8x8 1024 glb w4 pf nt3", 3, float, 16384, 4, 16, 16, 8, 8, 1, 1024, false, true, true, TW_GLOBAL, 4, true),
    };
#elif GROUP == 16  // ---- streaming-size non-power-of-two rows: 100k x 1000 (800 MB) ----
    const long long batch = 100000, outer = 1, inner = 1;
    const int N = 1000;
    std::vector<Variant> vs = {
        VN("10x10x10 t4 256 lds w1 nt0", 0, float, 1000, 3, 10, 10, 10, 1, 4, 256, false, true, true, TW_LDS, 1, false),
        VN("10x10x10 t4 256 lds w1 nt3", 3, float, 1000, 3, 10, 10, 10, 1, 4, 256, false, true, true, TW_LDS, 1, false),
        VN("10x10x10 t4 512 lds w2 nt3", 3, float, 1000, 3, 10, 10, 10, 1, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("10x10x10 t4 512 lds w2 nt0", 0, float, 1000, 3, 10, 10, 10, 1, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("5x5x5x8 t4 512 lds w2 nt3", 3, float, 1000, 4, 5, 5, 5, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("8x5x5x5 t4 512 lds w2 nt3", 3, float, 1000, 4, 8, 5, 5, 5, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("5x5x5x8 t4 512 lds w2 nt0", 0, float, 1000, 4, 5, 5, 5, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VN("10x10x10 t8 512 lds w2 nt3", 3, float, 1000, 3, 10, 10, 10, 1, 8, 512, false, true, true, TW_LDS, 2, false),
        VN("10x10x10 t4 256 lds w1 nt2", 2, float, 1000, 3, 10, 10, 10, 1, 4, 256, false, true, true, TW_LDS, 1, false),
        VN("10x10x10 t4 256 lds w1 nt1", 1, float, 1000, 3, 10, 10, 10, 1, 4, 256, false, true, true, TW_LDS, 1, false),
        VN("10x10x10 t2 256 lds w2 nt3", 3, float, 1000, 3, 10, 10, 10, 1, 2, 256, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 17  // ---- config 4 second pass again: fewer, larger passes (one LDS round trip) ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        V("c640 10x8x8 t16 512 lds pf", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 20x32 t16 512 lds", float, 640, 2, 20, 32, 1, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c640 20x32 t16 512 lds pf", float, 640, 2, 20, 32, 1, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 32x20 t16 512 lds", float, 640, 2, 32, 20, 1, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        V("c640 32x20 t16 512 lds pf", float, 640, 2, 32, 20, 1, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 20x32 t16 256 lds", float, 640, 2, 20, 32, 1, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        V("c640 32x20 t16 256 lds", float, 640, 2, 32, 20, 1, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        V("c640 20x32 t16 1024 lds", float, 640, 2, 20, 32, 1, 1, 16, 1024, true, true, true, TW_LDS, 1, false),
        V("c640 10x8x8 t16 256 glb w2", float, 640, 3, 10, 8, 8, 1, 16, 256, true, true, true, TW_GLOBAL, 2, false),
        V("c640 20x32 t16 256 glb w2", float, 640, 2, 20, 32, 1, 1, 16, 256, true, true, true, TW_GLOBAL, 2, false),
        V("c640 20x32 t8 256 lds w2", float, 640, 2, 20, 32, 1, 1, 8, 256, true, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 18  // ---- config 5 last pass: z axis of 10 x 128^3, columns of 128 with inner 16384 ----
    const long long batch = 10, outer = 1, inner = 16384;
    const int N = 128;
    std::vector<Variant> vs = {
        V("c128 8x4x4 t16 256 lds w4", float, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t16 128 lds w4", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_LDS, 4, false),
        V("c128 8x16 t16 128 lds w4", float, 128, 2, 8, 16, 1, 1, 16, 128, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t16 256 lds w4", float, 128, 2, 16, 8, 1, 1, 16, 256, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t32 256 lds w4", float, 128, 2, 16, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t32 512 lds w4", float, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 4, false),
        V("c128 8x4x4 t32 512 lds w4", float, 128, 3, 8, 4, 4, 1, 32, 512, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t64 512 lds w2", float, 128, 2, 16, 8, 1, 1, 64, 512, true, true, true, TW_LDS, 2, false),
        V("c128 16x8 t16 128 lds w4 pf", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_LDS, 4, true),
        V("c128 16x8 t32 256 lds w4 pf", float, 128, 2, 16, 8, 1, 1, 32, 256, true, true, true, TW_LDS, 4, true),
        V("c128 8x4x4 t16 256 lds w4 pf", float, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, true),
        V("c128 32x4 t16 64 lds w4", float, 128, 2, 32, 4, 1, 1, 16, 64, true, true, true, TW_LDS, 4, false),
        V("c128 16x8 t16 128 reg w4", float, 128, 2, 16, 8, 1, 1, 16, 128, true, true, true, TW_REG, 4, false),
        V("c128 16x8 t32 512 lds w4 pf", float, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 4, true),
        V("c128 16x8 t32 512 lds w2 pf", float, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 2, true),
        V("c128 8x16 t32 512 lds w4", float, 128, 2, 8, 16, 1, 1, 32, 512, true, true, true, TW_LDS, 4, false),
        V("c128 8x16 t32 512 lds w4 pf", float, 128, 2, 8, 16, 1, 1, 32, 512, true, true, true, TW_LDS, 4, true),
        V("c128 8x16 t64 512 lds w2 pf", float, 128, 2, 8, 16, 1, 1, 64, 512, true, true, true, TW_LDS, 2, true),
        V("c128 16x8 t64 1024 lds w4", float, 128, 2, 16, 8, 1, 1, 64, 1024, true, true, true, TW_LDS, 4, false),
        V("c128 4x4x8 t32 512 lds w4", float, 128, 3, 4, 4, 8, 1, 32, 512, true, true, true, TW_LDS, 4, false),
    };
#elif GROUP == 19  // ---- config 3 again: radix 3 first (248-byte runs in pass 0), radix 31 last ----
    const long long batch = 500000, outer = 1, inner = 1;
    const int N = 93;
    std::vector<Variant> vs = {
        VN("31x3 t64 192 w3 nt2 fd", 2, float, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("3x31 t64 192 w3 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("3x31 t64 192 w3 nt3 fd", 3, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, false, TW_LDS, 3, false),
        VN("3x31 t64 192 w2 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, false, TW_LDS, 2, false),
        VN("3x31 t64 192 w3 nt2 fd ld", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, true, TW_LDS, 3, false),
        VN("3x31 t64 192 w3 nt2 stg", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, false, false, TW_LDS, 3, false),
        VN("3x31 t32 96 w3 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 32, 96, false, true, false, TW_LDS, 3, false),
        VN("3x31 t128 384 w3 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 128, 384, false, true, false, TW_LDS, 3, false),
        VN("3x31 t64 256 w3 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 64, 256, false, true, false, TW_LDS, 3, false),
        VN("3x31 t64 192 w3 nt2 fd pf", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, false, TW_LDS, 3, true),
        VN("3x31 t64 192 w3 nt2 fd reg", 2, float, 93, 2, 3, 31, 1, 1, 64, 192, false, true, false, TW_REG, 3, false),
        VN("3x31 t48 192 w3 nt2 fd", 2, float, 93, 2, 3, 31, 1, 1, 48, 192, false, true, false, TW_LDS, 3, false),
    };
#elif GROUP == 20  // ---- config 4 second pass: wave-owned sub-problems (WSUB) ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        V("c640 10x8x8 t16 512 lds pf", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        V("c640 10x8x8 t16 640 lds pf", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
        VW("c640 10x8x8 t16 640 wsub pf", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
        VW("c640 10x8x8 t16 640 wsub", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, false),
        VW("c640 10x8x8 t16 320 wsub pf", float, 640, 3, 10, 8, 8, 1, 16, 320, true, true, true, TW_LDS, 1, true),
        VW("c640 8x8x10 t16 512 wsub pf", float, 640, 3, 8, 8, 10, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 8x8x10 t16 512 wsub", float, 640, 3, 8, 8, 10, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c640 8x10x8 t16 512 wsub pf", float, 640, 3, 8, 10, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 16x8x5 t16 1024 wsub", float, 640, 3, 16, 8, 5, 1, 16, 1024, true, true, true, TW_LDS, 1, false),
        VW("c640 16x8x5 t16 512 wsub pf", float, 640, 3, 16, 8, 5, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 16x5x8 t16 512 wsub pf", float, 640, 3, 16, 5, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 8x4x4x5 t16 512 wsub pf", float, 640, 4, 8, 4, 4, 5, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 8x5x4x4 t16 512 wsub pf", float, 640, 4, 8, 5, 4, 4, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 10x4x4x4 t16 640 wsub pf", float, 640, 4, 10, 4, 4, 4, 16, 640, true, true, true, TW_LDS, 1, true),
        // 80 KB of LDS per workgroup (twiddles from the global table): TWO workgroups per CU
        VW("c640 10x8x8 t16 640 wsub glb w5", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_GLOBAL, 5, false),
        VW("c640 10x8x8 t16 640 wsub glb w3", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_GLOBAL, 3, false),
        VW("c640 10x8x8 t16 640 wsub glb w5 pf", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_GLOBAL, 5, true),
        VW("c640 10x8x8 t16 320 wsub glb w3", float, 640, 3, 10, 8, 8, 1, 16, 320, true, true, true, TW_GLOBAL, 3, false),
        VW("c640 10x8x8 t16 320 wsub glb w3 pf", float, 640, 3, 10, 8, 8, 1, 16, 320, true, true, true, TW_GLOBAL, 3, true),
        V("c640 10x8x8 t16 512 glb w4", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_GLOBAL, 4, false),
    };
#elif GROUP == 21  // ---- four-step passes / long strided dims: columns of 1024 with WSUB ----
    const long long batch = 64, outer = 1, inner = 1024;
    const int N = 1024;
    std::vector<Variant> vs = {
        V("c1024 16x8x8 t16 512 lds", float, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c1024 16x8x8 t16 1024 wsub", float, 1024, 3, 16, 8, 8, 1, 16, 1024, true, true, true, TW_LDS, 1, false),
        VW("c1024 16x8x8 t16 512 wsub", float, 1024, 3, 16, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c1024 8x16x8 t16 512 wsub", float, 1024, 3, 8, 16, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c1024 8x8x16 t16 512 wsub", float, 1024, 3, 8, 8, 16, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c1024 16x4x4x4 t16 1024 wsub", float, 1024, 4, 16, 4, 4, 4, 16, 1024, true, true, true, TW_LDS, 1, false),
        VW("c1024 8x8x4x4 t16 512 wsub", float, 1024, 4, 8, 8, 4, 4, 16, 512, true, true, true, TW_LDS, 1, false),
        VW("c1024 8x4x4x8 t16 512 wsub", float, 1024, 4, 8, 4, 4, 8, 16, 512, true, true, true, TW_LDS, 1, false),
    };
#elif GROUP == 22  // ---- config 4 second pass: ping-pong halves sharing one LDS tile (tile_kernel_pp) ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        V("c640 10x8x8 t16 512 lds pf", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, true),
        VW("c640 10x8x8 t16 640 wsub pf", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
        VP("pp c640 10x8x8 t16 2x256", float, 640, 3, 10, 8, 8, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        VP("pp c640 10x8x8 t16 2x512", float, 640, 3, 10, 8, 8, 1, 16, 512, true, true, true, TW_LDS, 1, false),
        VP("pp c640 8x8x10 t16 2x256", float, 640, 3, 8, 8, 10, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        VP("pp c640 4x4x8x5 t16 2x256", float, 640, 4, 4, 4, 8, 5, 16, 256, true, true, true, TW_LDS, 1, false),
        VP("pp c640 4x4x8x5 t16 2x512", float, 640, 4, 4, 4, 8, 5, 16, 512, true, true, true, TW_LDS, 1, false),
        VP("pp c640 20x32 t16 2x256", float, 640, 2, 20, 32, 1, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        VP("pp c640 16x8x5 t16 2x256", float, 640, 3, 16, 8, 5, 1, 16, 256, true, true, true, TW_LDS, 1, false),
        VP("pp c640 10x8x8 t16 2x320", float, 640, 3, 10, 8, 8, 1, 16, 320, true, true, true, TW_LDS, 1, false),
    };
#elif GROUP == 23  // ---- 8K frame rows: 4320 x 7680 (fewer, larger passes?) ----
    const long long batch = 4320, outer = 1, inner = 1;
    const int N = 7680;
    std::vector<Variant> vs = {
        V("r7680 12x10x8x8 1024 lds pf", float, 7680, 4, 12, 10, 8, 8, 1, 1024, false, true, true, TW_LDS, 1, true),
        V("r7680 12x10x8x8 512 lds w2", float, 7680, 4, 12, 10, 8, 8, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r7680 12x10x8x8 512 lds w2 pf", float, 7680, 4, 12, 10, 8, 8, 1, 512, false, true, true, TW_LDS, 2, true),
        V("r7680 20x24x16 512 lds w2", float, 7680, 3, 20, 24, 16, 1, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r7680 20x24x16 512 lds w2 pf", float, 7680, 3, 20, 24, 16, 1, 1, 512, false, true, true, TW_LDS, 2, true),
        V("r7680 16x20x24 512 lds w2", float, 7680, 3, 16, 20, 24, 1, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r7680 24x20x16 512 glb w2", float, 7680, 3, 24, 20, 16, 1, 1, 512, false, true, true, TW_GLOBAL, 2, false),
        V("r7680 16x16x30 512 lds w2", float, 7680, 3, 16, 16, 30, 1, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r7680 32x16x15 512 lds w2", float, 7680, 3, 32, 16, 15, 1, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r7680 12x10x8x8 1024 glb w1 pf", float, 7680, 4, 12, 10, 8, 8, 1, 1024, false, true, true, TW_GLOBAL, 1, true),
    };
#elif GROUP == 24  // ---- real input (the reference benchmarks rfft): 100k x 1024 real -> complex ----
    const long long batch = 100000, outer = 1, inner = 1;
    const int N = 1024;
    std::vector<Variant> vs = {
        VR("r 16x8x8 t4 256 lds w4 nt0", 0, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VR("r 16x8x8 t4 256 lds w4 nt2", 2, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VR("r 16x8x8 t4 256 lds w4 nt3", 3, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VR("r 4x4x8x8 t4 512 lds w2 nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VR("r 4x4x8x8 t4 512 lds w2 nt2", 2, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 2, false),
        VR("r 8x8x16 t4 256 lds w4 nt3", 3, float, 1024, 3, 8, 8, 16, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VR("r 16x8x8 t4 256 lds w4 pf nt3", 3, float, 1024, 3, 16, 8, 8, 1, 4, 256, false, true, true, TW_LDS, 4, true),
        VR("r 16x8x8 t8 512 lds w2 nt3", 3, float, 1024, 3, 16, 8, 8, 1, 8, 512, false, true, true, TW_LDS, 2, false),
        VR("r 4x4x8x8 t4 512 lds w2 flat nt3", 3, float, 1024, 4, 4, 4, 8, 8, 4, 512, false, false, true, TW_LDS, 2, false),
        // round 3 (real input no longer conjugates its loads): two passes, other splits
        VR("r 32x32 t4 128 lds w2 nt3", 3, float, 1024, 2, 32, 32, 1, 1, 4, 128, false, true, true, TW_LDS, 2, false),
        VR("r 32x32 t8 256 lds w4 nt3", 3, float, 1024, 2, 32, 32, 1, 1, 8, 256, false, true, true, TW_LDS, 4, false),
        VR("r 16x16x4 t4 256 lds w4 nt3", 3, float, 1024, 3, 16, 16, 4, 1, 4, 256, false, true, true, TW_LDS, 4, false),
        VR("r 16x8x8 t2 128 lds w2 nt3", 3, float, 1024, 3, 16, 8, 8, 1, 2, 128, false, true, true, TW_LDS, 2, false),
        VR("r 16x8x8 t4 512 lds w4 nt3", 3, float, 1024, 3, 16, 8, 8, 1, 4, 512, false, true, true, TW_LDS, 4, false),
        VR("r 8x16x8 t4 256 lds w4 nt3", 3, float, 1024, 3, 8, 16, 8, 1, 4, 256, false, true, true, TW_LDS, 4, false),
    };
#elif GROUP == 25  // ---- 8K frame, other orientation: 7680 rows of 4320 ----
    const long long batch = 7680, outer = 1, inner = 1;
    const int N = 4320;
    std::vector<Variant> vs = {
        V("r4320 10x9x8x6 512 lds w1", float, 4320, 4, 10, 9, 8, 6, 1, 512, false, true, true, TW_LDS, 1, false),
        V("r4320 10x9x8x6 512 lds w2", float, 4320, 4, 10, 9, 8, 6, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r4320 10x9x8x6 512 lds w2 pf", float, 4320, 4, 10, 9, 8, 6, 1, 512, false, true, true, TW_LDS, 2, true),
        V("r4320 10x9x8x6 256 lds w2", float, 4320, 4, 10, 9, 8, 6, 1, 256, false, true, true, TW_LDS, 2, false),
        V("r4320 10x9x8x6 256 lds w3", float, 4320, 4, 10, 9, 8, 6, 1, 256, false, true, true, TW_LDS, 3, false),
        V("r4320 16x18x15 256 lds w2", float, 4320, 3, 16, 18, 15, 1, 1, 256, false, true, true, TW_LDS, 2, false),
        V("r4320 16x18x15 512 lds w2", float, 4320, 3, 16, 18, 15, 1, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r4320 18x16x15 320 lds w2", float, 4320, 3, 18, 16, 15, 1, 1, 320, false, true, true, TW_LDS, 2, false),
        V("r4320 12x10x6x6 512 lds w2", float, 4320, 4, 12, 10, 6, 6, 1, 512, false, true, true, TW_LDS, 2, false),
        V("r4320 10x9x8x6 512 glb w2", float, 4320, 4, 10, 9, 8, 6, 1, 512, false, true, true, TW_GLOBAL, 2, false),
        V("r4320 10x9x8x6 512 glb w4", float, 4320, 4, 10, 9, 8, 6, 1, 512, false, true, true, TW_GLOBAL, 4, false),
        V("r4320 10x9x8x6 t2 1024 glb w1", float, 4320, 4, 10, 9, 8, 6, 2, 1024, false, true, true, TW_GLOBAL, 1, false),
    };
#elif GROUP == 30  // ---- fp64: 100k x 1024 (1.6 GB each way) ----
    const long long batch = 100000, outer = 1, inner = 1;
    const int N = 1024;
    std::vector<Variant> vs = {
        VN("d 4x4x8x8 t2 256 lds w1 nt0", 0, double, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 1, false),
        VN("d 4x4x8x8 t2 256 lds w1 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 1, false),
        VN("d 4x4x8x8 t2 256 lds w2 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 2, false),
        VN("d 4x4x8x8 t4 512 lds w1 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 4, 512, false, true, true, TW_LDS, 1, false),
        VN("d 4x4x8x8 t2 512 lds w2 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 2, 512, false, true, true, TW_LDS, 2, false),
        VN("d 4x4x4x4x.. 8x8x4x4 t2 256 nt3", 3, double, 1024, 4, 8, 8, 4, 4, 2, 256, false, true, true, TW_LDS, 1, false),
        VN("d 8x8x16 t2 128 lds w2 nt3", 3, double, 1024, 3, 8, 8, 16, 1, 2, 128, false, true, true, TW_LDS, 2, false),
        VN("d 16x8x8 t2 128 lds w2 nt3", 3, double, 1024, 3, 16, 8, 8, 1, 2, 128, false, true, true, TW_LDS, 2, false),
        VN("d 4x4x8x8 t1 128 lds w4 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 1, 128, false, true, true, TW_LDS, 4, false),
        VN("d 4x4x8x8 t2 256 lds w3 nt3", 3, double, 1024, 4, 4, 4, 8, 8, 2, 256, false, true, true, TW_LDS, 3, false),
    };
#elif GROUP == 31  // ---- fp64: 500k x 93 ----
    const long long batch = 500000, outer = 1, inner = 1;
    const int N = 93;
    std::vector<Variant> vs = {
        VN("d93 31x3 t32 96 w1 flat-out nt0", 0, double, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
        VN("d93 31x3 t32 96 w1 flat-out nt2", 2, double, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
        VN("d93 31x3 t64 192 w1 flat-out nt2", 2, double, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 1, false),
        VN("d93 31x3 t64 192 w2 flat-out nt2", 2, double, 93, 2, 31, 3, 1, 1, 64, 192, false, true, false, TW_LDS, 2, false),
        VN("d93 31x3 t32 96 w2 flat-out nt2", 2, double, 93, 2, 31, 3, 1, 1, 32, 96, false, true, false, TW_LDS, 2, false),
        VN("d93 31x3 t21 64 w1 flat-out nt2", 2, double, 93, 2, 31, 3, 1, 1, 21, 64, false, true, false, TW_LDS, 1, false),
        VN("d93 3x31 t32 96 w1 nt2", 2, double, 93, 2, 3, 31, 1, 1, 32, 96, false, true, false, TW_LDS, 1, false),
        VN("d93 31x3 t32 96 w1 flat-both nt2", 2, double, 93, 2, 31, 3, 1, 1, 32, 96, false, false, false, TW_LDS, 1, false),
    };
#elif GROUP == 32  // ---- fp64: 500k x 128 ----
    const long long batch = 500000, outer = 1, inner = 1;
    const int N = 128;
    std::vector<Variant> vs = {
        VN("d128 8x4x4 t16 256 w1 nt0", 0, double, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 1, false),
        VN("d128 8x4x4 t16 256 w1 nt3", 3, double, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 1, false),
        VN("d128 8x4x4 t16 256 w2 nt3", 3, double, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_LDS, 2, false),
        VN("d128 8x4x4 t16 256 reg w2 nt3", 3, double, 128, 3, 8, 4, 4, 1, 16, 256, false, true, true, TW_REG, 2, false),
        VN("d128 8x16 t16 256 w2 nt3", 3, double, 128, 2, 8, 16, 1, 1, 16, 256, false, true, true, TW_LDS, 2, false),
        VN("d128 8x4x4 t8 128 w4 nt3", 3, double, 128, 3, 8, 4, 4, 1, 8, 128, false, true, true, TW_LDS, 4, false),
        VN("d128 8x4x4 t32 512 w1 nt3", 3, double, 128, 3, 8, 4, 4, 1, 32, 512, false, true, true, TW_LDS, 1, false),
    };
#elif GROUP == 33  // ---- fp64 config 4 second pass: columns of 640, inner 480, 100 images ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        V("dc640 4x4x8x5 t8 256 lds", double, 640, 4, 4, 4, 8, 5, 8, 256, true, true, true, TW_LDS, 1, false),
        V("dc640 4x4x8x5 t8 512 lds", double, 640, 4, 4, 4, 8, 5, 8, 512, true, true, true, TW_LDS, 1, false),
        V("dc640 4x4x8x5 t8 512 lds pf", double, 640, 4, 4, 4, 8, 5, 8, 512, true, true, true, TW_LDS, 1, true),
        V("dc640 10x8x8 t8 512 lds", double, 640, 3, 10, 8, 8, 1, 8, 512, true, true, true, TW_LDS, 1, false),
        V("dc640 10x8x8 t8 256 lds", double, 640, 3, 10, 8, 8, 1, 8, 256, true, true, true, TW_LDS, 1, false),
        VWD("dc640 10x8x8 t8 640 wsub", double, 640, 3, 10, 8, 8, 1, 8, 640, true, true, true, TW_LDS, 1, false),
        VWD("dc640 10x8x8 t8 640 wsub pf", double, 640, 3, 10, 8, 8, 1, 8, 640, true, true, true, TW_LDS, 1, true),
        VWD("dc640 10x8x8 t8 320 wsub pf", double, 640, 3, 10, 8, 8, 1, 8, 320, true, true, true, TW_LDS, 1, true),
        VWD("dc640 8x4x4x5 t8 512 wsub pf", double, 640, 4, 8, 4, 4, 5, 8, 512, true, true, true, TW_LDS, 1, true),
        VWD("dc640 8x4x4x5 t8 512 wsub", double, 640, 4, 8, 4, 4, 5, 8, 512, true, true, true, TW_LDS, 1, false),
        V("dc640 4x4x8x5 t4 256 lds w2", double, 640, 4, 4, 4, 8, 5, 4, 256, true, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 34  // ---- fp64 config 4 first pass: 64000 rows of 480 ----
    const long long batch = 64000, outer = 1, inner = 1;
    const int N = 480;
    std::vector<Variant> vs = {
        V("dr480 10x6x8 t4 128 lds w1", double, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 1, false),
        V("dr480 10x6x8 t4 128 lds w2", double, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 2, false),
        V("dr480 10x6x8 t4 128 lds w2 pf", double, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 2, true),
        V("dr480 10x6x8 t8 256 lds w1", double, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 1, false),
        V("dr480 10x6x8 t8 256 lds w1 pf", double, 480, 3, 10, 6, 8, 1, 8, 256, false, true, true, TW_LDS, 1, true),
        V("dr480 4x4x5x6 t4 256 lds w2", double, 480, 4, 4, 4, 5, 6, 4, 256, false, true, true, TW_LDS, 2, false),
        V("dr480 4x4x5x6 t8 512 lds w1", double, 480, 4, 4, 4, 5, 6, 8, 512, false, true, true, TW_LDS, 1, false),
        VN("dr480 10x6x8 t4 128 w2 nt1", 1, double, 480, 3, 10, 6, 8, 1, 4, 128, false, true, true, TW_LDS, 2, false),
        V("dr480 8x6x10 t4 128 lds w2", double, 480, 3, 8, 6, 10, 1, 4, 128, false, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 35  // ---- fp64 config 5: columns of 128 (inner 16384: z axis; the y axis has inner 128) ----
    const long long batch = 10, outer = 1, inner = 16384;
    const int N = 128;
    std::vector<Variant> vs = {
        V("dc128 8x4x4 t8 128 lds w1", double, 128, 3, 8, 4, 4, 1, 8, 128, true, true, true, TW_LDS, 1, false),
        V("dc128 8x4x4 t8 128 lds w4", double, 128, 3, 8, 4, 4, 1, 8, 128, true, true, true, TW_LDS, 4, false),
        V("dc128 8x16 t8 128 lds w4", double, 128, 2, 8, 16, 1, 1, 8, 128, true, true, true, TW_LDS, 4, false),
        V("dc128 16x8 t8 128 lds w4", double, 128, 2, 16, 8, 1, 1, 8, 128, true, true, true, TW_LDS, 4, false),
        V("dc128 16x8 t16 256 lds w4", double, 128, 2, 16, 8, 1, 1, 16, 256, true, true, true, TW_LDS, 4, false),
        V("dc128 8x4x4 t16 256 lds w4", double, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 4, false),
        V("dc128 8x4x4 t16 256 lds w2", double, 128, 3, 8, 4, 4, 1, 16, 256, true, true, true, TW_LDS, 2, false),
        V("dc128 16x8 t32 512 lds w2", double, 128, 2, 16, 8, 1, 1, 32, 512, true, true, true, TW_LDS, 2, false),
        V("dc128 8x4x4 t32 512 lds w2", double, 128, 3, 8, 4, 4, 1, 32, 512, true, true, true, TW_LDS, 2, false),
    };
#elif GROUP == 26  // ---- config 4 second pass once more: 8-column tiles (64-byte runs, 3 workgroups per CU) with everything round 2 added ----
    const long long batch = 100, outer = 1, inner = 480;
    const int N = 640;
    std::vector<Variant> vs = {
        VW("c640 10x8x8 t16 640 wsub pf", float, 640, 3, 10, 8, 8, 1, 16, 640, true, true, true, TW_LDS, 1, true),
        VW("c640 10x8x8 t8 320 wsub pf w3", float, 640, 3, 10, 8, 8, 1, 8, 320, true, true, true, TW_LDS, 3, true),
        VW("c640 10x8x8 t8 320 wsub w3", float, 640, 3, 10, 8, 8, 1, 8, 320, true, true, true, TW_LDS, 3, false),
        VW("c640 10x8x8 t8 640 wsub w1", float, 640, 3, 10, 8, 8, 1, 8, 640, true, true, true, TW_LDS, 1, false),
        V("c640 10x8x8 t8 256 lds w3", float, 640, 3, 10, 8, 8, 1, 8, 256, true, true, true, TW_LDS, 3, false),
        V("c640 10x8x8 t8 256 lds w3 pf", float, 640, 3, 10, 8, 8, 1, 8, 256, true, true, true, TW_LDS, 3, true),
        V("c640 10x8x8 t8 320 lds w3", float, 640, 3, 10, 8, 8, 1, 8, 320, true, true, true, TW_LDS, 3, false),
        V("c640 4x4x8x5 t8 256 lds w3", float, 640, 4, 4, 4, 8, 5, 8, 256, true, true, true, TW_LDS, 3, false),
        V("c640 4x4x8x5 t8 512 lds w2", float, 640, 4, 4, 4, 8, 5, 8, 512, true, true, true, TW_LDS, 2, false),
    };
#else
#error "define GROUP"
#endif

#if GROUP == 7 || GROUP == 13 || GROUP == 28 || GROUP == 29
    const size_t elems = (size_t)batch * outer * N * N;
#else
    const size_t elems = (size_t)batch * outer * inner * N;
#endif
    const size_t bytes = elems * 2 * sizeof(DT);
    std::vector<DT> h(elems * 2);
    unsigned s = 12345;
    for (auto& x : h) {
        s = s * 1664525u + 1013904223u;
        x = (DT)(((s >> 8) & 0xFFFF) / 65536.0f - 0.5f);
    }
    std::vector<DT> tw(2 * N);
    for (int n = 0; n < N; ++n) {
        tw[2 * n] = (DT)cos(-2.0 * M_PI * n / N);
        tw[2 * n + 1] = (DT)sin(-2.0 * M_PI * n / N);
    }
    void *din, *dout, *dref, *dtw;
    CK(hipMalloc(&din, bytes));
    CK(hipMalloc(&dout, bytes));
    CK(hipMalloc(&dref, bytes));
    CK(hipMalloc(&dtw, tw.size() * sizeof(DT)));
#ifdef MIFFT_STAMPS
    CK(hipMalloc(&g_stamps, 4096 * 16 * 8));
#endif
    CK(hipMemcpy(din, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(DT), hipMemcpyHostToDevice));

    const bool inplace = inner != 1;  // column kernels run in place on `out`
    auto prep = [&](void* dst) {
        if (inplace) CK(hipMemcpyAsync(dst, din, bytes, hipMemcpyDeviceToDevice, 0));
    };

    // correctness vs the first variant (sampled)
    const size_t sample = std::min<size_t>(elems * 2, 1u << 22);
    std::vector<DT> ref(sample), got(sample);
    prep(dref);
    vs[0].run(inplace ? dref : din, dref, dtw, batch, outer, inner);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), dref, sample * sizeof(DT), hipMemcpyDeviceToHost));
    // tail sample too
    std::vector<DT> ref_tail(sample), got_tail(sample);
    CK(hipMemcpy(ref_tail.data(), (char*)dref + bytes - sample * sizeof(DT), sample * sizeof(DT), hipMemcpyDeviceToHost));
    double refnorm = 0;
    for (DT x : ref) refnorm += (double)x * x;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> best(vs.size(), 1e30), sum(vs.size(), 0);
    std::vector<double> err(vs.size(), 0);
    for (size_t i = 0; i < vs.size(); ++i) {
        CK(hipMemsetAsync(dout, 0xFF, bytes, 0));
        prep(dout);
        vs[i].run(inplace ? dout : din, dout, dtw, batch, outer, inner);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, sample * sizeof(DT), hipMemcpyDeviceToHost));
        CK(hipMemcpy(got_tail.data(), (char*)dout + bytes - sample * sizeof(DT), sample * sizeof(DT), hipMemcpyDeviceToHost));
        double d = 0;
        for (size_t k = 0; k < sample; ++k) {
            double a = (double)got[k] - ref[k], b = (double)got_tail[k] - ref_tail[k];
            d += a * a + b * b;
        }
        err[i] = sqrt(d / (2 * refnorm));
    }
    const int rounds = 5, inner_iters = inplace ? 1 : 10;
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = 0; i < vs.size(); ++i) {
            if (inplace) {
                // in-place kernels: time single launches on fresh data (data stays finite either way)
                float tot = 0;
                for (int it = 0; it < 5; ++it) {
                    prep(dout);
                    CK(hipEventRecord(e0, 0));
                    vs[i].run(dout, dout, dtw, batch, outer, inner);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    tot += ms;
                }
                double ms = tot / 5;
                best[i] = std::min(best[i], ms);
                sum[i] += ms;
            } else {
                vs[i].run(din, dout, dtw, batch, outer, inner);  // warm
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < inner_iters; ++it) vs[i].run(din, dout, dtw, batch, outer, inner);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= inner_iters;
                best[i] = std::min(best[i], (double)ms);
                sum[i] += ms;
            }
        }
    }
#ifdef MIFFT_STAMPS
    {
        std::vector<unsigned long long> hs(4096 * 16);
        for (size_t i = 0; i < vs.size(); ++i) {
            CK(hipMemset(g_stamps, 0, 4096 * 16 * 8));
            prep(dout);
            vs[i].run(inplace ? dout : din, dout, dtw, batch, outer, inner);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hs.data(), g_stamps, 4096 * 16 * 8, hipMemcpyDeviceToHost));
            double ph[16] = {0};
            for (long long g = 0; g < g_last_grid; ++g)
                for (int k = 0; k < 16; ++k) ph[k] += (double)hs[g * 16 + k];
            printf("%-28s cycles per tile (thread 0):", vs[i].name.c_str());
            double tot = 0;
            for (int k = 0; k < 16; ++k) {
                printf(" [%d]%6.0f", k, ph[k] / (double)g_last_tiles);
                tot += ph[k] / (double)g_last_tiles;
            }
            printf("  sum %6.0f\n", tot);
        }
    }
#endif
    printf("GROUP %d  N=%d  tensor %.1f MB  (wg/cu override %d)\n", GROUP, N, bytes / 1e6, g_wg_override);
    printf("%-28s %9s %9s %8s %8s %10s\n", "variant", "min ms", "mean ms", "GB/s", "frac8T", "relerr");
    for (size_t i = 0; i < vs.size(); ++i) {
        double gbs = 2.0 * bytes / best[i] / 1e6;
        printf("%-28s %9.4f %9.4f %8.0f %8.3f %10.2e  lds %zu\n", vs[i].name.c_str(), best[i], sum[i] / rounds, gbs,
               gbs / 8000.0, err[i], vs[i].lds);
    }
    return 0;
}
