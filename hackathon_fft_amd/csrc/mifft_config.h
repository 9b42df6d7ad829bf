// mifft_config.h -- every size threshold and switch the plan-time policy depends on, in ONE place.
//
// PRODUCT build (libmifft.so): three environment variables, read once per process (listed in include/mifft.h):
//     MIFFT_JIT=0            no runtime specialisation (hipRTC); lengths without a precompiled kernel run on the literal stages
//     MIFFT_JIT_CACHE_DIR    directory of the on-disk cache of runtime-specialised code objects
//     MIFFT_JIT_VERBOSE=1    say on stderr why a runtime specialisation failed
// Every other number below is a constant derived from the size of the Infinity Cache.
//
// LAB build (libmifft_lab.so, -DMIFFT_EXPERIMENTAL [-DMIFFT_TESTING]): the measurement scripts under tools/ and the tests in
// tests/test_gpu_lab.py move the thresholds and switch the experimental kernels on through MIFFT_* variables, re-read at
// every plan creation (load_config() lists them).  None of that code is compiled into the product library.
#pragma once

#include <cstdlib>
#include <string>

namespace mifft {

// MI355X: 256 MiB memory-side cache between the eight L2s and HBM (/opt/skills/guides/MI355X_MICROARCH.md, "Infinity Cache").
// The store / load policies below are all statements about what fits it.
constexpr double kInfinityCacheBytes = 256.0 * 1024.0 * 1024.0;

struct Config {
    // ---- product switches ----
    bool jit = true;
    bool jit_verbose = false;
    std::string jit_cache_dir;

    // ---- thresholds, in bytes moved per exec (read + write) unless noted; measured on MI355X, DESIGN.md 3.1c / 3.5 ----
    // batched 1-D transforms whose tensors dwarf the cache: streaming twins (non-temporal loads and stores)
    double streaming_min_bytes = 2.235 * kInfinityCacheBytes;  // 0.60 GB
    // x or out alone would fit, together they thrash: non-temporal STORES only
    double nts_min_bytes = 0.931 * kInfinityCacheBytes;  // 0.25 GB
    double nts_max_bytes = 2.049 * kInfinityCacheBytes;  // 0.55 GB
    // N = 93 (flat-copy store) gains from non-temporal stores much earlier
    double nts_small_min_bytes = 0.186 * kInfinityCacheBytes;  // 0.05 GB
    // N-D plans: `out` (bytes of the OUTPUT tensor) fits the cache but x + out do not: the first pass reads x non-temporally
    double nd_out_min_bytes = 0.596 * kInfinityCacheBytes;  // 160 MB
    double nd_out_max_bytes = 0.931 * kInfinityCacheBytes;  // 250 MB
    // ... the wave-private 128 x 128 plane prefetches far enough ahead that the hint pays only from here
    double nd_plane128_min_bytes = 0.782 * kInfinityCacheBytes;  // 210 MB
    // big batches of >= 16384-point rows prefer the two-launch four-step over one workgroup per row
    long long fourstep_min_n = 16384;
    double fourstep_min_bytes = 0.119 * kInfinityCacheBytes;  // 32 MB of output

    // prime radices above 32 run in LDS: as Rader's cyclic convolution when R - 1 splits into register butterflies and
    // R >= rader_min (cooperative conjugate-pair pass otherwise).  Per 128 MB, cooperative -> Rader (tools/big_prime_probe2.py):
    // R = 37 0.082 -> 0.069 ms, 97 0.152 -> 0.078, 131 0.200 -> 0.108, 251 0.307 -> 0.124, 1009 1.06 -> 0.17, 4093 4.26 -> 0.30;
    // a length with TWO such primes (1517 = 41 * 37: 0.151 -> 0.186) keeps the cooperative pass below rader_min_pair.
    int rader_min = 33;
    int rader_min_pair = 128;
    // primes whose R - 1 has a prime factor above 31 (83, 509, 2039 ...): zero-padded convolution of a smooth length >=
    // 2 R - 3 in a scratch block, from this size on (per 128 MB, cooperative -> padded: 179 0.235 -> 0.229, 263 0.33 -> 0.25,
    // 509 0.57 -> 0.26, 1019 1.11 -> 0.32, 2039 2.16 -> 0.48; below, the cooperative pass wins: 83 0.142 against 0.295)
    int rader_pad_min = 200;

    // ---- Infinity-Cache policy of N-D schedules: bit 0 non-temporal first pass, bit 1 alternating walk ----
    int nd_mode = 3;

    // ---- lab switches (constant in the product build) ----
    long long fs_strided_min_n = 0;  // > 0: strided dimensions from this length try the two-pass four-step first
    bool fourstep_strided = true;    // false: long strided dimensions take the transposed route
    long long fs_n1 = 0;             // forced first factor of the strided four-step
    bool row2d = true;               // four-step inside one LDS plane for 8192 / 16384-point rows
    int jit_nt = -1;                 // -1 default policy, 0 no non-temporal runtime-specialised kernels, 1 streaming hints too
    bool jit_image = false;          // L2-resident image kernel (negative result, DESIGN_EXPERIMENTS.md)
    bool dpp = false;                // wave-shuffle radix-3 kernel for N = 93 (negative result)
    std::string jit_cols_cfg;        // MIFFT_JIT_COLS_CFG="10x12x16:8:512": radices : tile : threads of the runtime-specialised column
                                     // tiles of that length (tools/cols_cfg_sweep.py)
    std::string jit_rows_cfg;        // MIFFT_JIT_ROWS_CFG="16x15:16:256:1": radices : tile : threads : prefetch of the runtime-specialised
                                     // ROW tiles of that length (packed real rows: of N / 2)
    bool skip_gen_table = false;     // MIFFT_SKIP_GEN_TABLE=1: generated table entries ignored (runtime-specialised kernels instead; A/B)
    bool herm_first_axis = true;     // half-spectrum schedule of three-pass real-input plans: halve the FIRST pass's dimension
                                     // (MIFFT_HERM_FIRST_AXIS=0: only the pass before the last stores half; A/B baseline)
    bool r2c_rows = false;           // packed real rows (TileCfg::R2C) as the half-store first pass: MIFFT_R2C=1.  Measured: within
                                     // -5 % / +5 % of the tuned half-store row kernels (DESIGN_EXPERIMENTS.md R3.8) -- not a default
    bool half_store = true;          // half-store kernel in front of a Hermitian last pass (MIFFT_HS=0: A/B baseline)
    int herm = 1;                    // Hermitian twins for the last pass of real-input N-D plans: 1 = where herm_pays() says so,
                                     // MIFFT_HERM=0 never (A/B baseline), MIFFT_HERM=2 wherever a twin exists (small test shapes)
    int grid_per_cu = 0;             // forces the persistent grid of every table kernel (tools/grid_sweep.py)
    std::string jit_defines;         // extra -D options for the runtime compiler (A/B of kernel-header macros)
    bool test_fail_scratch_alloc = false;  // -DMIFFT_TESTING: the next scratch allocation reports out-of-memory
};

inline Config load_config() {
    Config c;
    auto env = [](const char* name) -> const char* {
        const char* v = getenv(name);
        return (v && v[0]) ? v : nullptr;
    };
    if (const char* v = env("MIFFT_JIT")) c.jit = v[0] != '0';
    if (const char* v = env("MIFFT_JIT_VERBOSE")) c.jit_verbose = v[0] == '1';
    if (const char* v = env("MIFFT_JIT_CACHE_DIR")) c.jit_cache_dir = v;
#ifdef MIFFT_EXPERIMENTAL
    if (const char* v = env("MIFFT_ND_CACHE")) c.nd_mode = atoi(v);
    if (const char* v = env("MIFFT_ND_CACHE_MAX_MB")) c.nd_out_max_bytes = atof(v) * 1e6;
    if (const char* v = env("MIFFT_ND_CACHE_MIN_MB")) c.nd_out_min_bytes = atof(v) * 1e6;
    if (const char* v = env("MIFFT_NTS_MIN_BYTES")) c.nts_min_bytes = atof(v);
    if (const char* v = env("MIFFT_NTS_MAX_BYTES")) c.nts_max_bytes = atof(v);
    if (const char* v = env("MIFFT_FOURSTEP_MIN_N")) c.fourstep_min_n = atoll(v);
    if (const char* v = env("MIFFT_FS_STRIDED_MIN_N")) c.fs_strided_min_n = atoll(v);
    if (const char* v = env("MIFFT_FOURSTEP_STRIDED")) c.fourstep_strided = v[0] != '0';
    if (const char* v = env("MIFFT_FS_N1")) c.fs_n1 = atoll(v);
    if (const char* v = env("MIFFT_ROW2D")) c.row2d = v[0] != '0';
    if (const char* v = env("MIFFT_JIT_NT")) c.jit_nt = atoi(v);
    if (const char* v = env("MIFFT_JIT_IMAGE")) c.jit_image = v[0] == '1';
    if (const char* v = env("MIFFT_DPP")) c.dpp = v[0] == '1';
    if (const char* v = env("MIFFT_JIT_COLS_CFG")) c.jit_cols_cfg = v;
    if (const char* v = env("MIFFT_JIT_ROWS_CFG")) c.jit_rows_cfg = v;
    if (const char* v = env("MIFFT_SKIP_GEN_TABLE")) c.skip_gen_table = v[0] == '1';
    if (const char* v = env("MIFFT_HERM_FIRST_AXIS")) c.herm_first_axis = v[0] != '0';
    if (const char* v = env("MIFFT_R2C")) c.r2c_rows = v[0] == '1';
    if (const char* v = env("MIFFT_HS")) c.half_store = v[0] != '0';
    if (const char* v = env("MIFFT_HERM")) c.herm = v[0] == '0' ? 0 : v[0] == '2' ? 2 : 1;
    if (const char* v = env("MIFFT_GRID_PER_CU")) c.grid_per_cu = atoi(v);
    if (const char* v = env("MIFFT_JIT_DEFINES")) c.jit_defines = v;
    if (const char* v = env("MIFFT_RADER_MIN")) c.rader_min = c.rader_min_pair = atoi(v);
    if (const char* v = env("MIFFT_RADER_PAD_MIN")) c.rader_pad_min = atoi(v);
#endif
#ifdef MIFFT_TESTING
    if (const char* v = env("MIFFT_TEST_FAIL_SCRATCH_ALLOC")) c.test_fail_scratch_alloc = v[0] == '1';
#endif
    return c;
}

// product: read once; lab: refreshed at every plan creation (config_refresh) so that a script can move a knob between plans
inline Config& config_storage() {
    static Config c = load_config();
    return c;
}
inline const Config& config() { return config_storage(); }
inline void config_refresh() {
#if defined(MIFFT_EXPERIMENTAL) || defined(MIFFT_TESTING)
    config_storage() = load_config();
#endif
}

}  // namespace mifft
