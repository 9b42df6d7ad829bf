// mifft_api.cpp -- the C ABI of libmifft (see include/mifft.h).
//
// Host scheduler replacing _run_gpu_nd_fft (fft/fft/_ndim_fft_gpu.mojo:462-642):
// the contiguous (last) dimension is transformed first, reading `x` and writing
// `out`; every earlier dimension is then transformed IN PLACE on `out` by a
// strided tile kernel.  The reference's transpose launches and scratch buffer
// do not exist here: d launches instead of d + 2(d-1).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "mifft_config.h"
#include "mifft_internal.h"

namespace mifft {

static thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_error(hipError_t e, const char* what) {
    g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return MIFFT_ERR_HIP;
}

size_t Plan::in_elem_bytes() const { return dtype_size(in_dtype) * (size_t)in_components; }
size_t Plan::out_elem_bytes() const { return dtype_size(out_dtype) * 2; }

// W_N^n = exp(-+ 2*pi*i*n/N), evaluated in long double and rounded once to the
// output dtype (reference formula: fft/fft/_utils.mojo:63-104, which rounds theta
// to the working dtype first; see DESIGN.md "numerics").
template <typename T>
static void fill_twiddles(std::vector<T>& tab, int64_t N, bool inverse) {
    tab.resize((size_t)2 * N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int64_t n = 0; n < N; ++n) {
        // exact octant values where they exist
        long double re, im;
        const int64_t n8 = 8 * n;
        if (n8 % N == 0 && ((n8 / N) % 2) == 0) {
            switch ((n8 / N) / 2) {
                case 0: re = 1, im = 0; break;
                case 1: re = 0, im = -1; break;
                case 2: re = -1, im = 0; break;
                default: re = 0, im = 1; break;
            }
        } else {
            long double th = -two_pi * (long double)n / (long double)N;
            re = cosl(th);
            im = sinl(th);
        }
        if (inverse) im = -im;
        tab[2 * n] = (T)re;
        tab[2 * n + 1] = (T)im;
    }
}

static void free_plan_device(Plan& p) {
    for (auto& ps : p.passes) {
        if (ps.d_twiddle) (void)hipFree(ps.d_twiddle);
        if (ps.d_aux) (void)hipFree(ps.d_aux);
        if (ps.d_aux2) (void)hipFree(ps.d_aux2);
        if (ps.d_aux3) (void)hipFree(ps.d_aux3);
        ps.d_twiddle = ps.d_aux = ps.d_aux2 = ps.d_aux3 = nullptr;
    }
    if (p.d_scratch) (void)hipFree(p.d_scratch);
    p.d_scratch = nullptr;
}

hipError_t upload_twiddle_table(int out_dtype, int64_t N, bool inverse, void** d_table) {
    hipError_t err;
    if (out_dtype == MIFFT_F32) {
        std::vector<float> tab;
        fill_twiddles(tab, N, inverse);
        err = hipMalloc(d_table, tab.size() * sizeof(float));
        if (err == hipSuccess) err = hipMemcpy(*d_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    } else {
        std::vector<double> tab;
        fill_twiddles(tab, N, inverse);
        err = hipMalloc(d_table, tab.size() * sizeof(double));
        if (err == hipSuccess) err = hipMemcpy(*d_table, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
    }
    return err;
}

// Switches to the plan's device for the duration of a call and restores the caller's current device
// afterwards (the library must not leave a side effect on the host framework's device state).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

static int device_count_quiet() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // namespace mifft

using namespace mifft;

// The library is built with -fvisibility=hidden: ONLY the C ABI below is exported.  (Its internal C++ names live in namespace
// mifft; exported, they would interpose with equally named symbols of the program that loads the library.)
#pragma GCC visibility push(default)
extern "C" {

const char* mifft_last_error(void) { return g_last_error.c_str(); }

const char* mifft_status_string(int s) {
    switch (s) {
        case MIFFT_OK: return "ok";
        case MIFFT_ERR_BAD_RANK: return "bad rank";
        case MIFFT_ERR_BAD_DIM: return "bad dimension";
        case MIFFT_ERR_BAD_COMPONENTS: return "bad component count";
        case MIFFT_ERR_BAD_DTYPE: return "bad dtype";
        case MIFFT_ERR_BAD_BASES: return "bases do not factor the length";
        case MIFFT_ERR_BASE_ONE: return "base 1";
        case MIFFT_ERR_NO_BASES: return "no bases";
        case MIFFT_ERR_BAD_BATCH: return "bad batch";
        case MIFFT_ERR_TOO_LARGE: return "dimension too large";
        case MIFFT_ERR_NO_DEVICE: return "no device";
        case MIFFT_ERR_HIP: return "hip error";
        case MIFFT_ERR_NULL: return "null argument";
        case MIFFT_ERR_ALIAS: return "x and out overlap";
        case MIFFT_ERR_BUFFER_TOO_SMALL: return "buffer too small";
    }
    return "unknown";
}

int mifft_version(void) { return MIFFT_VERSION_MAJOR * 100 + MIFFT_VERSION_MINOR; }

int mifft_device_count(void) { return device_count_quiet(); }

int mifft_ordered_bases(uint32_t length, const uint32_t* bases, int nbases, uint32_t* ordered_out, int capacity) {
    std::vector<uint64_t> user;
    for (int i = 0; i < nbases; ++i) user.push_back(bases[i]);
    std::vector<uint32_t> ord, proc;
    std::string err;
    int rc = plan_ordered_bases(length, user, ord, proc, err);
    if (rc) return set_error(rc, err);
    for (size_t i = 0; i < ord.size() && (int)i < capacity; ++i) ordered_out[i] = ord[i];
    return (int)ord.size();
}

int mifft_estimate_bases(uint32_t length, int target_gpu, uint32_t* bases_out, int capacity) {
    if (length < 2) return set_error(MIFFT_ERR_BAD_DIM, "length must be >= 2");
    auto b = plan_estimate_bases(length, target_gpu != 0);
    for (size_t i = 0; i < b.size() && (int)i < capacity; ++i) bases_out[i] = (uint32_t)b[i];
    return (int)b.size();
}

int mifft_plan_create(mifft_plan** out_plan, int device, int in_dtype, int out_dtype, int ndim,
                      const int64_t* dims, int64_t batch, int in_components, int inverse,
                      const uint32_t* bases_flat, const int32_t* bases_len, uint32_t flags) {
    return mifft_plan_create_slab(out_plan, device, in_dtype, out_dtype, ndim, dims, batch, in_components, inverse,
                                  bases_flat, bases_len, flags, 0);
}

int mifft_plan_create_slab(mifft_plan** out_plan, int device, int in_dtype, int out_dtype, int ndim,
                           const int64_t* dims, int64_t batch, int in_components, int inverse,
                           const uint32_t* bases_flat, const int32_t* bases_len, uint32_t flags,
                           int64_t whole_batch) {
    if (!out_plan) return set_error(MIFFT_ERR_NULL, "out_plan is NULL");
    if (whole_batch != 0 && whole_batch < batch) {
        *out_plan = nullptr;
        return set_error(MIFFT_ERR_BAD_BATCH, "whole_batch must be 0 or >= batch");
    }
    *out_plan = nullptr;
    // ---- validation = _check_layout_conditions_nd (fft/fft/fft.mojo:20-46) ----
    if (ndim < 1 || ndim > MIFFT_MAX_DIMS)
        return set_error(MIFFT_ERR_BAD_RANK, "The rank should be bigger than 2 (1..6 transformed dims supported)");
    if (!dims) return set_error(MIFFT_ERR_NULL, "dims is NULL");
    if (in_components < 1 || in_components > 2)
        return set_error(MIFFT_ERR_BAD_COMPONENTS, "The last dimension of in_layout should be 1 or 2");
    if (out_dtype != MIFFT_F32 && out_dtype != MIFFT_F64)
        return set_error(MIFFT_ERR_BAD_DTYPE, "out_dtype must be floating point");
    if (dtype_size(in_dtype) == 0) return set_error(MIFFT_ERR_BAD_DTYPE, "unsupported in_dtype");
    if (batch < 0) return set_error(MIFFT_ERR_BAD_BATCH, "batch must be >= 0");
    for (int i = 0; i < ndim; ++i)
        if (dims[i] < 2) return set_error(MIFFT_ERR_BAD_DIM, "no inner dimension should be of size 1");
    if ((bases_flat == nullptr) != (bases_len == nullptr))
        return set_error(MIFFT_ERR_NULL, "bases_flat and bases_len must both be given or both be NULL");

    mifft_plan* h = new mifft_plan();
    Plan& p = h->p;
    p.device = device;
    p.in_dtype = in_dtype;
    p.out_dtype = out_dtype;
    p.ndim = ndim;
    p.batch = batch;
    p.sel_batch = whole_batch;
    p.in_components = in_components;
    p.inverse = inverse ? 1 : 0;
    p.flags = flags;
    p.prod = 1;
    for (int i = 0; i < ndim; ++i) {
        p.dims[i] = dims[i];
        p.prod *= dims[i];
    }

    // ---- radix planning per dimension (host only; errors before any device use) ----
    std::vector<std::vector<uint32_t>> ordered(ndim), processed(ndim);
    const uint32_t* bp = bases_flat;
    for (int i = 0; i < ndim; ++i) {
        std::vector<uint64_t> user;
        if (bases_flat) {
            if (bases_len[i] < 0) {
                delete h;
                return set_error(MIFFT_ERR_NO_BASES, "negative bases_len");
            }
            for (int k = 0; k < bases_len[i]; ++k) user.push_back(*bp++);
        } else {
            user = plan_estimate_bases((uint64_t)dims[i], /*gpu_target=*/true);
        }
        std::string err;
        int rc = plan_ordered_bases((uint64_t)dims[i], user, ordered[i], processed[i], err);
        if (rc) {
            delete h;
            return set_error(rc, err);
        }
    }

    // ---- 32-bit lane offsets of the column tiles (tile_kernel.h, lane_off): a strided dimension must span fewer than
    //      2^32 elements, N * stride < 2^32 (a single transform of more than 32 GB; checked before any device work) ----
    {
        double inner = 1.0;
        for (int i = ndim - 1; i >= 0; --i) {
            if (i < ndim - 1 && (double)dims[i] * inner + 64.0 >= 4294967296.0) {
                delete h;
                return set_error(MIFFT_ERR_TOO_LARGE,
                                 "dimension " + std::to_string(i) + " (" + std::to_string(dims[i]) + " points at stride " +
                                     std::to_string((long long)inner) + ") spans 2^32 elements or more: strided tiles "
                                     "address their elements with 32-bit lane offsets");
            }
            inner *= (double)dims[i];
        }
    }

    // ---- device ----
    const int ndev = device_count_quiet();
    if (device < 0 || device >= ndev) {
        delete h;
        return set_error(MIFFT_ERR_NO_DEVICE,
                         "libmifft has no CPU path: device " + std::to_string(device) + " is not a usable HIP device (" +
                             std::to_string(ndev) + " visible)");
    }
    DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) {
        delete h;
        return hip_error(e, "hipSetDevice");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) p.num_cus = prop.multiProcessorCount;

    // ---- Infinity-Cache policy for N-D transforms (thresholds: mifft_config.h).  Bit 0 of nd_mode = the first pass reads x
    //      with non-temporal loads when `out` fits the 256-MiB cache, so that x does not displace the row results the next
    //      pass reads (100 x 640 x 480: rows 96.2 -> 92.6 us, columns 102.9 -> 88.1 us); bit 1 = in-place passes walk their
    //      tiles in alternating directions, starting with the lines written last (10 x 128^3: 0.131 -> 0.128 ms,
    //      100 x 64^3 with both: 0.151 -> 0.139 ms).  Results are bit-identical in every mode. ----
    config_refresh();  // (lab builds re-read their MIFFT_* switches per plan; a no-op in the product library)
    const Config& cfg = config();
    const int nd_mode = cfg.nd_mode;
    {
        const double out_bytes = p.size_batch() * (double)p.prod * (double)p.out_elem_bytes();
        // ... and only when x and out together do NOT fit: below ~160 MB per tensor everything stays cache-resident
        // between the passes (and between execs), and non-temporal loads of x cost 6-9 % (100-image batches of
        // 640 x 480 at 59 / 118 MB: 0.0510 -> 0.0551 / 0.0834 -> 0.0905 ms, 64^3 the same; tools/nd_size_probe.py)
        p.cache_resident_nd = ndim >= 2 && (nd_mode & 1) && out_bytes <= cfg.nd_out_max_bytes && out_bytes >= cfg.nd_out_min_bytes;
    }

    // ---- passes in execution order: last dimension first ----
    p.stage_radices = ordered;
    auto upload_twiddles = [&](DimPass& ps) -> hipError_t {
        // (four-step rows inside LDS: the table of the row side, M / N1 points; the column side's goes to d_aux when it
        //  differs, the M-entry table to d_aux3)
        if (ps.r2c) {  // packed real rows: the passes run N / 2 points; the unpacking needs W_N^k, forward
            hipError_t e2 = upload_twiddle_table(out_dtype, ps.N / 2, inverse != 0, &ps.d_twiddle);
            if (e2 == hipSuccess) e2 = upload_twiddle_table(out_dtype, ps.N, false, &ps.d_aux);
            return e2;
        }
        hipError_t e = upload_twiddle_table(out_dtype, ps.row2d_m > 0 ? ps.row2d_m / ps.N1 : ps.N, inverse != 0, &ps.d_twiddle);
        if (e == hipSuccess && ps.plane_needs_tw1) e = upload_twiddle_table(out_dtype, ps.N1, inverse != 0, &ps.d_aux);
        if (e == hipSuccess && ps.row2d_m > 0) e = upload_twiddle_table(out_dtype, ps.row2d_m, false, &ps.d_aux3);
        if (e == hipSuccess && ps.needs_counters) {  // 16 per-launch counters + the sticky error word
            e = hipMalloc(&ps.d_aux2, 20 * sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(ps.d_aux2, 0, 20 * sizeof(unsigned));
        }
        return e;
    };
    // Will the LAST pass (dimension 0) be a Hermitian twin?  Asked when the pass over dimension 1 is selected: that pass may
    // then store only the lower half of its dimension (TileCfg::HS).  The same selection the loop below makes for i == 0,
    // on a scratch DimPass (whatever it uploads is freed again); the loop checks at the end that both agree.
    int herm_known = -1;
    auto herm_possible = [&]() -> bool {
        return cfg.herm != 0 && cfg.half_store && ndim >= 2 && ndim <= 4 && in_components == 1 && !(flags & MIFFT_FLAG_FAITHFUL_STAGES);
    };
    // (asked ONCE, after p.hs_selected has been set: a half-store pass in front widens the cases in which the twin pays)
    auto last_pass_will_be_hermitian = [&]() -> bool {
        if (herm_known >= 0) return herm_known != 0;
        herm_known = 0;
        if (!herm_possible()) return false;
        DimPass t;
        t.dim_index = 0;
        t.N = dims[0];
        t.inner = 1;
        for (int k = 1; k < ndim; ++k) t.inner *= dims[k];
        t.outer = 1;
        t.radices = ordered[0];
        t.processed = processed[0];
        t.first = false;
        if (t.inner >= (1ll << 30)) return false;
#ifdef MIFFT_EXPERIMENTAL  // (lab knob that sends long strided dimensions to the four-step before anything else)
        if (cfg.fs_strided_min_n > 0 && t.N >= cfg.fs_strided_min_n) return false;
#endif
        t.want_herm = t.herm_only = true;
        bool ok = select_fast(p, t);
        t.herm_only = false;
        if (!ok && t.N <= 4096) {
            std::string why;
            ok = select_jit(p, t, why) && t.herm_d2 > 0;
        }
        if (t.d_aux) (void)hipFree(t.d_aux);
        if (t.d_aux2) (void)hipFree(t.d_aux2);
        if (t.d_aux3) (void)hipFree(t.d_aux3);
        herm_known = ok ? 1 : 0;
        return ok;
    };
    // Half-spectrum schedule of a real-input plan with exactly one pass between the first and the last (rows / columns /
    // columns of a 3-D plan, plane / columns / columns of a 4-D one): the FIRST pass stores only the lower half of the dimension
    // it transforms last (dims[2]), the middle pass transforms only those columns, the Hermitian last pass halves that same
    // dimension (Plan::herm_axis = 2) -- every pass but the last moves half the tensor.  Falls back to the schedule with the
    // half store on the pass before the last (herm_axis = 1) when a kernel is missing.
    bool allow_first_axis = cfg.herm_first_axis, first_axis = false;
    auto try_first_axis = [&](DimPass& t, bool found) -> bool {  // t: the first pass with a half-store kernel selected
        if (!found) return false;
        p.herm_axis = 2;
        p.hs_selected = true;
        herm_known = -1;
        if (last_pass_will_be_hermitian()) return true;
        p.herm_axis = 1;
        p.hs_selected = false;
        herm_known = -1;  // (asked again, for the other schedule, when the pass over dimension 1 is selected)
        if (t.d_aux) (void)hipFree(t.d_aux);
        if (t.d_aux2) (void)hipFree(t.d_aux2);
        t.d_aux = t.d_aux2 = nullptr;
        return false;
    };
build_passes:
    for (int i = ndim - 1; i >= 0; --i) {
        DimPass ps;
        ps.dim_index = i;
        ps.N = dims[i];
        ps.inner = 1;
        for (int k = i + 1; k < ndim; ++k) ps.inner *= dims[k];
        ps.outer = 1;
        for (int k = 0; k < i; ++k) ps.outer *= dims[k];
        ps.radices = ordered[i];
        ps.processed = processed[i];
        ps.first = i == ndim - 1;
        // the LAST pass of a real-input 2-D .. 4-D plan may exploit the Hermitian symmetry of what it reads (TileCfg::HERM):
        // half the reads and butterflies, every result stored at (k, c) and conjugated at (-k, -c)
        ps.want_herm = cfg.herm != 0 && i == 0 && ndim >= 2 && ndim <= 4 && in_components == 1 && !(flags & MIFFT_FLAG_FAITHFUL_STAGES) &&
                       ps.inner < (1ll << 30);
        bool ok = false;
        if (!(flags & MIFFT_FLAG_FAITHFUL_STAGES)) {
            if (i == ndim - 1 && ndim >= 2) {  // try to fuse the two innermost dimensions in one LDS plane
                DimPass pl = ps;
                pl.dim_index2 = i - 1;
                pl.N1 = dims[i - 1];
                pl.outer = 1;
                for (int k = 0; k < i - 1; ++k) pl.outer *= dims[k];
                std::string whyp;
                bool fused = false;
                if (allow_first_axis && ndim == 4 && herm_possible()) {  // plane / columns / columns: halve the plane's column side
                    DimPass t = pl;
                    t.want_half = true;
                    t.store_lim = (int)(dims[2] / 2);
                    if (try_first_axis(t, select_fast_plane(p, t) || select_jit_plane(p, t, whyp))) {
                        pl = t;
                        fused = first_axis = true;
                    }
                }
                if (!fused && i - 1 == 1 && herm_possible()) {  // the plane's column side is dimension 1: half store
                    DimPass t = pl;
                    t.want_half = true;
                    t.store_lim = (int)(dims[1] / 2);
                    p.hs_selected = select_fast_plane(p, t) || select_jit_plane(p, t, whyp);
                    fused = p.hs_selected && last_pass_will_be_hermitian();
                    if (fused) pl = t;
                    p.hs_selected = fused;
                }
                if (!fused) fused = select_fast_plane(p, pl) || select_jit_plane(p, pl, whyp);
#ifdef MIFFT_EXPERIMENTAL  // L2-resident image kernel: a documented negative result, lab builds only
                if (!fused) fused = select_jit_image(p, pl, whyp);
#endif
                if (fused) {
                    ps = pl;
                    ok = true;
                    --i;  // dimension i-1 is covered by this pass
                }
            }
            // rows / columns / columns of a real-input 3-D plan: the row pass stores the lower half of every row
            if (!ok && allow_first_axis && ndim == 3 && i == 2 && herm_possible()) {
                DimPass t = ps;
                t.want_half = true;
                t.store_lim = (int)(dims[2] / 2);
                bool found = false;
#ifdef MIFFT_EXPERIMENTAL  // packed real rows: a measured tie with the tuned half-store kernels, lab builds only
                std::string whyr;
                found = cfg.r2c_rows && select_jit_r2c(p, t, whyr);
#endif
                if (!found) found = select_fast(p, t);
                if (!found && ps.N <= 4096) {
                    DimPass u = ps;
                    const bool tuned = select_fast(p, u) && u.regime_twin;
                    std::string whyh;
                    found = !tuned && select_jit(p, t, whyh);
                }
                if (try_first_axis(t, found)) {
                    ps = t;
                    ok = first_axis = true;
                }
            }
            if (first_axis && i == 1) {  // the middle pass: only the columns the first pass stored
                ps.col_prefix = dims[2] / 2 + 1;
                for (int k = 3; k < ndim; ++k) ps.col_prefix *= dims[k];
            }
            // the pass over dimension 1 of a plan whose last pass will be a Hermitian twin: a half-store kernel, tuned or
            // runtime specialised, before anything else
            // (a kernel tuned for a size regime -- a non-temporal twin -- is not traded for a runtime-specialised one: 64 x 1024^2
            //  0.372 -> 0.422 ms with `rows1024_16x8x8_hs_r_jit` in place of `rows1024_16x8x8_r_nt`)
            if (!ok && i == 1 && !first_axis && herm_possible()) {
                DimPass t = ps;
                t.want_half = true;
                t.store_lim = (int)(dims[1] / 2);
                bool found = false;
#ifdef MIFFT_EXPERIMENTAL  // (2-D plans: this is the row pass)
                std::string whyr;
                found = cfg.r2c_rows && ps.first && select_jit_r2c(p, t, whyr);
#endif
                if (!found) found = select_fast(p, t);
                if (!found && ps.N <= 4096) {
                    DimPass u = ps;
                    const bool tuned = select_fast(p, u) && u.regime_twin;
                    std::string whyh;
                    found = !tuned && select_jit(p, t, whyh);
                }
                p.hs_selected = found;
                if (found && last_pass_will_be_hermitian()) {
                    ps = t;
                    ok = true;
                } else {
                    if (found) {  // (nothing but Rader tables could have been uploaded for t)
                        if (t.d_aux) (void)hipFree(t.d_aux);
                        if (t.d_aux2) (void)hipFree(t.d_aux2);
                    }
                    p.hs_selected = false;
                }
            }
            if (!ok) ok = select_row2d(p, ps);  // 16384-point rows: four-step inside one LDS plane, one launch
            // Long rows of a big batched 1-D transform: two column-tile passes (four-step) move the tensor twice
            // at ~4.7 TB/s each (256 MB: 0.22 ms), which beats one workgroup per 128-KiB row (one workgroup per CU,
            // 2 TB/s: 0.26 ms) once the tensor fills the GPU in both passes; 8192-point rows are still faster in
            // one kernel (0.13 ms).  (Config::fourstep_min_n; 0 = never prefer the four-step.)
            if (!ok && ndim == 1 && p.in_components == 2 && p.in_dtype == p.out_dtype) {
                const long long min_n = cfg.fourstep_min_n;
                const double bytes = p.size_batch() * (double)ps.N * (double)p.out_elem_bytes();
                if (min_n > 0 && ps.N >= min_n && bytes >= cfg.fourstep_min_bytes) {
                    std::string why4;
                    const size_t before = p.passes.size();
                    if (build_fourstep(p, i, why4) && p.passes.size() == before + 2) continue;
                    if (p.alloc_failed) {
                        free_plan_device(p);
                        delete h;
                        return set_error(MIFFT_ERR_HIP, "four-step: " + why4);
                    }
                    // no two-pass split: drop whatever was appended and fall through to the single-kernel path
                    for (size_t k = before; k < p.passes.size(); ++k) {
                        DimPass& q = p.passes[k];
                        if (q.d_twiddle) (void)hipFree(q.d_twiddle);
                        if (q.d_aux) (void)hipFree(q.d_aux);
                        if (q.d_aux2) (void)hipFree(q.d_aux2);
                        if (q.d_aux3) (void)hipFree(q.d_aux3);
                    }
                    p.passes.resize(before);
                    if (p.d_scratch) {
                        (void)hipFree(p.d_scratch);
                        p.d_scratch = nullptr;
                        p.scratch_bytes = 0;
                    }
                }
            }
#ifdef MIFFT_EXPERIMENTAL
            // lab knob: strided dimensions of at least Config::fs_strided_min_n points try the two-pass four-step before the
            // single column tile (whose tile narrows to 8 / 4 columns = 64- / 32-byte runs beyond ~1300 / 2600 points)
            if (!ok && ps.inner != 1 && cfg.fs_strided_min_n > 0 && (long long)ps.N >= cfg.fs_strided_min_n) {
                std::string whyt;
                if (build_fourstep_strided(p, i, whyt)) continue;
                if (p.alloc_failed) {
                    free_plan_device(p);
                    delete h;
                    return set_error(MIFFT_ERR_HIP, "four-step of strided dimension " + std::to_string(i) + ": " + whyt);
                }
            }
            if (!ok) {
                std::string whys;
                ok = select_jit_streaming_rows(p, ps, whys);  // streaming hints on runtime-specialised rows (negative result)
            }
            if (!ok) ok = select_dpp_rows(p, ps);  // wave-shuffle radix 3 for N = 93 (negative result)
#endif
            // last pass of a real-input 2-D .. 4-D plan: a Hermitian twin (half the columns) beats any full column kernel --
            // a tuned twin first, else the runtime-specialised one, and only then the ordinary tables
            if (!ok && ps.want_herm) {
                ps.herm_only = true;
                ok = select_fast(p, ps);
                ps.herm_only = false;
                if (!ok && ps.N <= 4096) {
                    std::string whyh;
                    ok = select_jit(p, ps, whyh) && ps.herm_d2 > 0;
                }
            }
            if (!ok) ok = select_fast(p, ps);
            // no table entry: specialise the tile kernel for this length now (strided dimensions up to 4096 points;
            // longer ones are better off on the transposed route below)
            if (!ok && (ps.inner == 1 || ps.N <= 4096)) {
                std::string whyj;
                ok = select_jit(p, ps, whyj);
            }
            // long strided dimension without a fused column tile: transposes + the contiguous-row kernel beat the
            // literal-stage column fallback by an order of magnitude (and lift its 10 240-point limit)
            if (!ok && ps.inner != 1 && ps.N > 4096) {
                std::string whyt;
                // two column passes through the scratch (four-step) when both factors have fused tiles ...
                if (build_fourstep_strided(p, i, whyt)) continue;
                if (p.alloc_failed) {
                    free_plan_device(p);
                    delete h;
                    return set_error(MIFFT_ERR_HIP, "four-step of strided dimension " + std::to_string(i) + ": " + whyt);
                }
                // ... else the reference's own route: transpose, contiguous-row kernel, transpose back
                if (build_transposed_dim(p, i, ordered[i], processed[i], whyt)) continue;
                if (p.alloc_failed) {  // out of device memory is an error, not a reason to try a slower kernel
                    free_plan_device(p);
                    delete h;
                    return set_error(MIFFT_ERR_HIP, "transposed route of dimension " + std::to_string(i) + ": " + whyt);
                }
            }
        }
        if (!ok) {
            std::string why;
            ok = select_generic(p, ps, why);
            if (!ok) {
                // too long for one workgroup's LDS: four-step over two column-tile passes (contiguous dim of a
                // batched 1-D complex transform; the reference has no path at all here off NVIDIA clusters)
                std::string why4;
                if (i == ndim - 1 && build_fourstep(p, i, why4)) continue;
                const bool oom = p.alloc_failed;
                free_plan_device(p);
                delete h;
                if (oom) return set_error(MIFFT_ERR_HIP, "four-step: " + why4);
                return set_error(MIFFT_ERR_TOO_LARGE, why + (why4.empty() ? "" : "; four-step: " + why4));
            }
        }
        if (ps.prepare) {
            int prc = ps.prepare();
            if (prc) {
                free_plan_device(p);
                delete h;
                return prc;
            }
        }
        e = upload_twiddles(ps);
        p.passes.push_back(ps);
        if (e != hipSuccess) {
            free_plan_device(p);
            delete h;
            return hip_error(e, "twiddle table upload");
        }
    }
    if (first_axis) {  // the half-spectrum schedule needs all three of its kernels; else build the plan again without it
        const bool good = p.passes.size() == 3 && p.passes[0].hs && p.passes[1].prefix_ok && p.passes[2].herm_d2 > 0 &&
                          p.passes[2].herm_dj == (int)dims[2];
        if (!good) {
            free_plan_device(p);
            p.passes.clear();
            p.scratch_bytes = 0;
            p.alloc_failed = false;
            p.herm_axis = 1;
            p.hs_selected = false;
            herm_known = -1;
            allow_first_axis = first_axis = false;
            goto build_passes;
        }
    }
    {   // a half-store pass is only right in front of a Hermitian last pass (both selections are deterministic; this is the
        // net under them)
        bool any_hs = false;
        for (const DimPass& q : p.passes) any_hs = any_hs || q.hs;
        if (any_hs && !(p.passes.back().herm_d2 > 0)) {
            free_plan_device(p);
            delete h;
            return set_error(MIFFT_ERR_HIP, "internal: half-store pass without a Hermitian last pass");
        }
    }
    if (nd_mode & 2) {
        int k = 0;
        for (DimPass& ps : p.passes) {
            const bool in_place = (ps.src_buf < 0 && !ps.first) || (ps.src_buf == 1 && ps.dst_buf == 1);
            if (in_place) ps.reverse = (k++ % 2) == 0;
        }
    }
    *out_plan = h;
    return MIFFT_OK;
}

int mifft_exec_batch(const mifft_plan* plan, const void* x, void* out, int64_t first, int64_t count,
                     void* stream) {
    if (!plan) return set_error(MIFFT_ERR_NULL, "plan is NULL");
    const Plan& p = plan->p;
    if (first < 0 || count < 0 || first + count > p.batch)
        return set_error(MIFFT_ERR_BAD_BATCH, "batch range out of bounds");
    if (count == 0) return MIFFT_OK;
    if (!x || !out) return set_error(MIFFT_ERR_NULL, "x or out is NULL");
    const size_t in_row = (size_t)p.prod * p.in_elem_bytes(), out_row = (size_t)p.prod * p.out_elem_bytes();
    const char* xb = (const char*)x + (size_t)first * in_row;
    char* ob = (char*)out + (size_t)first * out_row;
    // out-of-place contract (reference: first stage reads x, all writes go elsewhere)
    if (xb < ob + (size_t)count * out_row && ob < xb + (size_t)count * in_row)
        return set_error(MIFFT_ERR_ALIAS, "x and out must not overlap");
    DeviceGuard guard(p.device);
    MIFFT_HIP_TRY(guard.err);
    hipStream_t s = (hipStream_t)stream;
    char* sb = p.d_scratch ? (char*)p.d_scratch + (size_t)first * out_row : nullptr;
    auto buf = [&](int which) -> char* { return which == 0 ? (char*)xb : which == 2 ? sb : ob; };
    for (const DimPass& ps : p.passes) {
        const void* src = ps.src_buf >= 0 ? buf(ps.src_buf) : (ps.first ? (const char*)xb : ob);
        void* dst = ps.dst_buf >= 0 ? buf(ps.dst_buf) : ob;
        int rc = ps.launch(p, ps, src, dst, count, s);
        if (rc) return rc;
    }
    return MIFFT_OK;
}

int mifft_exec(const mifft_plan* plan, const void* x, void* out, void* stream) {
    if (!plan) return set_error(MIFFT_ERR_NULL, "plan is NULL");
    return mifft_exec_batch(plan, x, out, 0, plan->p.batch, stream);
}

void mifft_plan_destroy(mifft_plan* plan) {
    if (!plan) return;
    {
        DeviceGuard guard(plan->p.device);
        free_plan_device(plan->p);
    }
    delete plan;
}

int mifft_plan_stages(const mifft_plan* plan, int dim, uint32_t* radices_out, int capacity) {
    if (!plan) return set_error(MIFFT_ERR_NULL, "plan is NULL");
    if (dim < 0 || dim >= plan->p.ndim) return set_error(MIFFT_ERR_BAD_RANK, "dim out of range");
    const std::vector<uint32_t>& r = plan->p.stage_radices[dim];
    for (size_t i = 0; i < r.size() && (int)i < capacity; ++i) radices_out[i] = r[i];
    return (int)r.size();
}

const char* mifft_plan_kernel_name(const mifft_plan* plan, int dim) {
    if (!plan) return "";
    for (const DimPass& ps : plan->p.passes)
        if (ps.dim_index == dim || ps.dim_index2 == dim) return ps.kernel_name;
    return "";
}

int mifft_plan_num_launches(const mifft_plan* plan) {
    if (!plan) return set_error(MIFFT_ERR_NULL, "plan is NULL");
    return (int)plan->p.passes.size();
}

size_t mifft_plan_in_bytes(const mifft_plan* plan) {
    return plan ? (size_t)plan->p.batch * plan->p.prod * plan->p.in_elem_bytes() : 0;
}

size_t mifft_plan_out_bytes(const mifft_plan* plan) {
    return plan ? (size_t)plan->p.batch * plan->p.prod * plan->p.out_elem_bytes() : 0;
}

size_t mifft_plan_scratch_bytes(const mifft_plan* plan) { return plan && plan->p.d_scratch ? plan->p.scratch_bytes : 0; }

int mifft_plan_device_status(const mifft_plan* plan, void* stream, uint32_t* flags_out) {
    if (!plan || !flags_out) return set_error(MIFFT_ERR_NULL, "plan or flags_out is NULL");
    *flags_out = 0;
    DeviceGuard guard(plan->p.device);
    MIFFT_HIP_TRY(guard.err);
    for (const DimPass& ps : plan->p.passes) {
        if (!ps.needs_counters || !ps.d_aux2) continue;
        hipStream_t s = (hipStream_t)stream;
        unsigned v = 0;
        MIFFT_HIP_TRY(hipMemcpyAsync(&v, (const unsigned*)ps.d_aux2 + 16, sizeof v, hipMemcpyDeviceToHost, s));
        MIFFT_HIP_TRY(hipMemsetAsync((unsigned*)ps.d_aux2 + 16, 0, sizeof v, s));
        MIFFT_HIP_TRY(hipStreamSynchronize(s));
        *flags_out |= v;
    }
    return MIFFT_OK;
}

int mifft_time_exec(const mifft_plan* plan, const void* x, void* out, void* stream, int iters, float* ms_out) {
    if (!plan || !ms_out) return set_error(MIFFT_ERR_NULL, "plan or ms_out is NULL");
    if (iters < 1) iters = 1;
    DeviceGuard guard(plan->p.device);
    MIFFT_HIP_TRY(guard.err);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    int rc = MIFFT_OK;
    if (he == hipSuccess) he = hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == MIFFT_OK && he == hipSuccess; ++i) rc = mifft_exec(plan, x, out, stream);
    if (he == hipSuccess) he = hipEventRecord(e1, s);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (he != hipSuccess) return hip_error(he, "mifft_time_exec events");
    *ms_out = ms / (float)iters;
    return MIFFT_OK;
}

int mifft_jit_precompile(int in_dtype, int out_dtype, int64_t length, int strided, int real_input,
                         size_t* code_bytes_out) {
    if (dtype_size(in_dtype) == 0) return set_error(MIFFT_ERR_BAD_DTYPE, "unknown in dtype");
    if (out_dtype != MIFFT_F32 && out_dtype != MIFFT_F64) return set_error(MIFFT_ERR_BAD_DTYPE, "out dtype must be f32 or f64");
    if (length < 2) return set_error(MIFFT_ERR_BAD_DIM, "length must be >= 2");
    std::string why;
    const int rc = jit_precompile(in_dtype, out_dtype, length, strided, real_input, code_bytes_out, why);
    if (rc != MIFFT_OK) return set_error(rc, why);
    return MIFFT_OK;
}

}  // extern "C"
#pragma GCC visibility pop
