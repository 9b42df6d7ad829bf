// mifft_internal.h -- plan structures shared by the C-ABI layer and the kernel
// launchers of libmifft (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mifft.h"
#include "mifft_config.h"

namespace mifft {

// ---- host-only planner (planner.cpp) --------------------------------------
// Independent implementation of the reference's radix planning rules:
//   _times_divisible_by / _build_ordered_bases   fft/fft/_utils.mojo:125-183
//   _get_ordered_bases_processed_list            fft/fft/_utils.mojo:186-221
//   _estimate_best_bases                         fft/fft/fft.mojo:49-104
int plan_ordered_bases(uint64_t length, const std::vector<uint64_t>& user, std::vector<uint32_t>& ordered,
                       std::vector<uint32_t>& processed, std::string& err);
std::vector<uint64_t> plan_estimate_bases(uint64_t length, bool gpu_target);

// ---- error plumbing (mifft_api.cpp) ----------------------------------------
int set_error(int code, const std::string& msg);
int hip_error(hipError_t e, const char* what);
#define MIFFT_HIP_TRY(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ::mifft::hip_error(_e, #expr); \
    } while (0)

// ---- plan -------------------------------------------------------------------
struct Plan;
struct DimPass;

// Launches the 1-D transforms of one dimension for `count` leading-batch
// entries.  `in` is x (only for the last dimension, which runs first) or `out`
// itself (in-place tile passes for the strided dimensions).
typedef int (*LaunchFn)(const Plan& plan, const DimPass& pass, const void* in, void* out, int64_t count,
                        hipStream_t stream);

struct DimPass {
    int dim_index = 0;
    int dim_index2 = -1;  // >= 0: a fused plane pass that also transforms this (next-outer) dimension
    int64_t row2d_m = 0;  // > 0: four-step rows inside LDS (plane_kernel_wp<.., FS>): M-entry forward table in d_aux3
    int64_t N1 = 0;       // length of dim_index2
    int64_t N = 0;        // transform length
    int64_t inner = 1;    // element stride of this dim = prod(dims after it)
    int64_t outer = 1;    // prod(dims before it)   (times batch at launch)
    std::vector<uint32_t> radices;    // descending stage radices (reference order)
    std::vector<uint32_t> processed;  // prefix products P_b
    void* d_twiddle = nullptr;        // device table W_N^n, n in [0,N), complex<out dtype>
    void* d_aux = nullptr;            // kernel-family specific device table (may be null)
    const char* kernel_name = "none";
    LaunchFn launch = nullptr;
    int (*prepare)() = nullptr;  // plan-time, on the plan's device (per-device kernel attributes)
    // geometry chosen by the kernel family at plan time
    int tile = 1;          // transforms per workgroup tile
    int ld = 0;            // LDS leading dimension (complex elements)
    int threads = 256;
    size_t lds_bytes = 0;
    bool first = false;    // reads x (with in_dtype / in_components) instead of out
    // buffer routing of the four-step passes (kernels_fourstep.hip): 0 = x, 1 = out, 2 = plan scratch;
    // -1 = the default (first ? x : out) -> out
    int src_buf = -1, dst_buf = -1;
    int64_t fs_n1 = 0, fs_n2 = 0;  // four-step factors of the dimension (transpose + twiddle pass)
    void* d_aux2 = nullptr;
    bool needs_counters = false;      // image kernel: d_aux2 holds 16 unsigned counters, zeroed before every launch
    bool plane_needs_tw1 = false;     // rectangular fused plane: d_aux holds the W_N1 table of the column side
    void* jit_fn = nullptr;           // runtime-compiled kernel (hipFunction_t) of kernels_jit.cpp passes
    void* d_aux3 = nullptr;           // TSTORE passes: [tile][N] table of W^(c*k1) for the columns of one tile
    bool reverse = false;             // in-place pass that walks its tiles last-to-first (TileParams::reverse)
    int wg_per_cu = 0;                // workgroups per CU of the persistent grid (0 = the LDS / wave-count formula)
    // last pass of a real-input 2-D .. 4-D plan: the kernel families may take a Hermitian twin (TileCfg::HERM) that reads
    // and transforms only the columns up to their mirror.  want_herm is set by the scheduler, herm_d0 / herm_d1 / herm_d2 (the
    // trailing dimensions of the column space) by the family that selected such a kernel (0 = an ordinary kernel).
    bool want_herm = false;
    bool herm_only = false;  // (select_fast: accept only Hermitian twins -- the scheduler tries those first)
    int herm_d0 = 0, herm_d1 = 0, herm_d2 = 0;
    // ... and the half axis of the column space (TileParams::herm_dj ...): size, stride, row length, covered prefix of a row
    int herm_dj = 0, herm_js = 0, herm_L = 0, herm_H = 0;
    // an ordinary column pass between a half-store first pass and the Hermitian last pass transforms only the first
    // col_prefix columns of its column space (0 = all): the others were not stored and are not read
    long long col_prefix = 0;
    bool prefix_ok = false;  // the selected kernel honours col_prefix (the column tiles do, special routes do not)
    // the pass BEFORE such a last pass (it transforms dims[1], the dimension the Hermitian pass halves; a plane: its column
    // side) may store only the results 0 .. store_lim = dims[1] / 2 (TileCfg::HS): want_half asks the families for such a
    // kernel ONLY, `hs` says the selected kernel is one
    bool want_half = false, hs = false;
    bool r2c = false;  // packed real rows (TileCfg::R2C): the kernel runs N / 2 points; d_twiddle is that table, d_aux W_N^k
    int store_lim = 0;
    bool regime_twin = false;  // select_fast took an entry tuned for a size regime (non-temporal twin): not to be traded for
                               // a runtime-specialised half-store kernel
};

struct Plan {
    int device = 0;
    int in_dtype = MIFFT_F32, out_dtype = MIFFT_F32;
    int ndim = 1;
    int64_t dims[MIFFT_MAX_DIMS] = {};
    int64_t batch = 0;
    // mifft_plan_create_slab: the plan is one slab of a larger batch; every size-dependent choice (streaming twins, store
    // policy, cache policy, four-step threshold) is made for THIS many transforms, so that the slab's results equal the
    // same rows of one plan over the whole batch bit for bit.  0 = the plan's own batch.
    int64_t sel_batch = 0;
    double size_batch() const { return (double)(sel_batch > 0 ? sel_batch : batch); }
    int64_t prod = 1;
    int in_components = 2;
    int inverse = 0;
    uint32_t flags = 0;
    int num_cus = 256;
    std::vector<DimPass> passes;  // in execution order: last dim first
    std::vector<std::vector<uint32_t>> stage_radices;  // per dim: the user's ordered stages (introspection)
    void* d_scratch = nullptr;  // four-step only: one tensor of the output size (the reference's calc_buf,
                                // fft/fft/_ndim_fft_gpu.mojo:185, exists for EVERY plan; here only for dims > 16384)
    size_t scratch_bytes = 0;
    bool alloc_failed = false;  // a route builder ran out of device memory: plan creation reports MIFFT_ERR_HIP
    // N-D transform whose `out` tensor fits the 256-MiB Infinity Cache: the first pass reads x with non-temporal loads
    // (x does not displace the row results) and later in-place passes alternate their walking direction
    bool cache_resident_nd = false;
    // plan creation only: the pass over dimension 1 has been given a half-store kernel (TileCfg::HS), which widens the cases
    // in which a Hermitian last pass pays (herm_pays)
    bool hs_selected = false;
    // plan creation only: the dimension the Hermitian last pass halves (index into dims).  1: the dimension of the pass right
    // before it, which then stores only half of its results; 2: the dimension the FIRST pass transforms -- the rows of a
    // three-pass 3-D plan, the column side of the plane of a 4-D one -- with a half-store first pass and a middle pass over the
    // lower half of its columns: every pass but the last then moves half the tensor.
    int herm_axis = 1;
    size_t in_elem_bytes() const;
    size_t out_elem_bytes() const;  // bytes of one complex output element
};

// Hermitian twins: the trailing dimensions of the column space of the last pass (dims[1..ndim-1], at most three; absent
// leading ones are 1), and the flat prefix of columns that are not beyond their own mirror image
inline void herm_set_dims(const Plan& plan, DimPass& pass) {
    const int n = plan.ndim, a = plan.herm_axis;
    pass.herm_d2 = (int)plan.dims[n - 1];
    pass.herm_d1 = n >= 3 ? (int)plan.dims[n - 2] : 1;
    pass.herm_d0 = n >= 4 ? (int)plan.dims[n - 3] : 1;
    long long js = 1;
    for (int k = a + 1; k < n; ++k) js *= plan.dims[k];
    pass.herm_dj = (int)plan.dims[a];
    pass.herm_js = (int)js;
    pass.herm_L = (int)(plan.dims[a] * js);
    pass.herm_H = (int)((plan.dims[a] / 2 + 1) * js);
}
// tiles of `tile` columns that cover the transformed part of one image's column space
inline long long herm_tiles_per_outer(const DimPass& pass, int tile) {
    const long long rows = pass.inner / pass.herm_L, tpr = (pass.herm_H + tile - 1) / tile;
    return rows * tpr;
}

// Does a Hermitian twin (TileCfg::HERM) with `tile` columns per tile pay for this pass?  Measured with tools/herm_probe.py
// (DESIGN_EXPERIMENTS.md R3.6), whole transform against the ordinary kernels:
//   * tiles of whole 128-byte lines store their mirrored lines whole (contiguous runs with a carried column): 5-11 % at
//     every size when the rows of the trailing dimension are a whole number of tiles; ragged rows leave every mirrored line
//     in pieces (6 x 360^3 +3 %) unless the tensor stays in the Infinity Cache, where the pieces merge;
//   * narrower tiles (strided dimensions beyond ~1024 points) store every mirrored line in pieces: alone a gain (3-15 %) only
//     while the tensor is cache resident (with the table's own three-pass configurations: 4K frames 7-16 %);
//   * with a half-store pass in front (plan.hs_selected: that pass skips half its writes) the ragged and the 4- / 8-column cases
//     pay beyond the cache too: 40 x 1920 x 1080 0.98 -> 0.91, 6 x 360^3 1.03 -> 0.93, 300 x 600 x 500 1.02 -> 0.95;
//   * a Hermitian tile stores twice what an ordinary one does, so half the tiles must also mean clearly fewer ROUNDS of the
//     persistent grid: 1 x 64^4 (2112 tiles on 1024 workgroups: 3 rounds instead of 4) runs 12 % SLOWER, 1 x 256^3 (the
//     same 3 : 4, but four times the arithmetic per stored line) 5 % faster, 4 x 64^4 (9 : 16) 6 % faster.
inline bool herm_pays(const Plan& plan, const DimPass& pass, int tile, size_t lds_bytes, int threads) {
    if (config().herm == 2) return true;  // (lab build only: forced)
    const double out_bytes = plan.size_batch() * (double)plan.prod * (double)plan.out_elem_bytes();
    const bool resident = out_bytes <= (double)kInfinityCacheBytes;
    const long long d2 = plan.dims[plan.ndim - 1];
    const long long run_bytes = (long long)tile * (long long)plan.out_elem_bytes();
    if (run_bytes >= 128) {
        if (d2 % tile != 0 && !resident && !plan.hs_selected) return false;
    } else if (run_bytes < 32 || !(resident || plan.hs_selected)) {
        return false;
    }
    DimPass dims = pass;
    herm_set_dims(plan, dims);
    long long per_cu = (160 * 1024) / (long long)(lds_bytes ? lds_bytes : 1);  // the persistent grid, as tile_grid<>
    per_cu = std::max<long long>(1, std::min<long long>(std::min<long long>(per_cu, 2048 / (threads > 0 ? threads : 256)), 16));
    const long long grid = (long long)plan.num_cus * per_cu;
    const long long images = (long long)plan.size_batch() * pass.outer;
    const long long full = images * ((pass.inner + tile - 1) / tile);
    const long long half = images * herm_tiles_per_outer(dims, tile);
    const long long rounds_full = (full + grid - 1) / grid, rounds_half = (half + grid - 1) / grid;
    return (double)rounds_half <= (pass.N >= 128 ? 0.75 : 0.70) * (double)rounds_full;
}

// kernel families; each returns true and fills pass.launch/tile/... when it
// accepts the (plan, pass) pair.
bool select_generic(const Plan& plan, DimPass& pass, std::string& why_not);
bool select_fast(const Plan& plan, DimPass& pass);
// LAB BUILD ONLY (-DMIFFT_EXPERIMENTAL): wave-autonomous rows of N = 3 * R0 points: radix R0 in registers, radix 3 across
// three lanes by DPP (kernels_dpp.hip)
bool select_dpp_rows(const Plan& plan, DimPass& pass);
// the tile kernel specialised at plan time with hipRTC for a length without a table entry (kernels_jit.cpp)
bool select_jit(const Plan& plan, DimPass& pass, std::string& why_not);
// first pass over a REAL tensor whose last pass will be a Hermitian twin: rows read as N / 2 packed complex points, unpacked
// into the half spectrum by the store loop (TileCfg::R2C); pass.want_half asks for it.  LAB BUILD ONLY.
bool select_jit_r2c(const Plan& plan, DimPass& pass, std::string& why_not);
// four-step helpers: the transposed + twiddled column pass (reads x: real / integer input allowed) and cheap
// feasibility predicates for scoring factorisations without compiling
bool select_jit_streaming_rows(const Plan& plan, DimPass& pass, std::string& why_not);  // LAB BUILD ONLY
// the non-temporal-store window of batched 1-D transforms (bytes moved per exec, kernels_fast.hip; Config::nts_*)
bool nts_window(const Plan& plan, double total_bytes);
// contiguous dimension of N1 * N2 points as a four-step inside one LDS plane (kernels_fast.hip)
bool select_row2d(const Plan& plan, DimPass& pass);
bool nts_window_bytes(double total_bytes);  // the same window without the one-dimension condition (single-pass planes)
bool select_jit_plane(const Plan& plan, DimPass& pass, std::string& why_not);
// LAB BUILD ONLY: the two innermost dimensions of images that fit one XCD's L2: rows, XCD-local barrier, columns from L2
bool select_jit_image(const Plan& plan, DimPass& pass, std::string& why_not);
bool select_jit_tstore(const Plan& plan, DimPass& pass, std::string& why_not);
// first four-step pass of a strided dimension (TileCfg::FS1): pass.N = N1, pass.fs_n2 = N2
bool select_jit_fs1(const Plan& plan, DimPass& pass, std::string& why_not);
// a strided dimension beyond one column tile as two column passes through the plan scratch (kernels_fourstep.hip)
bool build_fourstep_strided(Plan& plan, int dim_index, std::string& why_not);
bool jit_tstore_feasible(const Plan& plan, int64_t n1, int64_t n2);
bool jit_cols_feasible(const Plan& plan, int64_t n, int64_t inner);
int jit_precompile(int in_dtype, int out_dtype, int64_t n, int cols, int in_real, size_t* code_bytes, std::string& why);
// fused pass over the two innermost dimensions (pass.N = contiguous dim, pass.N1 = the next one)
bool select_fast_plane(const Plan& plan, DimPass& pass);
// a contiguous dimension too long for one workgroup: three passes (column FFTs of N1, transpose + twiddle,
// column FFTs of N2).  Appends its passes to plan.passes and allocates plan.d_scratch.
bool build_fourstep(Plan& plan, int dim_index, std::string& why_not);
// a strided dimension too long for an in-place column tile: transpose -> row kernel -> transpose through the scratch
bool build_transposed_dim(Plan& plan, int dim_index, const std::vector<uint32_t>& radices,
                          const std::vector<uint32_t>& processed, std::string& why_not);
hipError_t upload_twiddle_table(int out_dtype, int64_t N, bool inverse, void** d_table);

inline size_t dtype_size(int dt) {
    switch (dt) {
        case MIFFT_F32: return 4;
        case MIFFT_F64: return 8;
        case MIFFT_U8: return 1;
        case MIFFT_I32: return 4;
        case MIFFT_I8: return 1;
        case MIFFT_I16: return 2;
        case MIFFT_U16: return 2;
        case MIFFT_F16: return 2;
        case MIFFT_BF16: return 2;
    }
    return 0;
}

}  // namespace mifft

// the opaque C handle
struct mifft_plan {
    mifft::Plan p;
};
