// mifft.hpp -- C++ host mirror of the reference's GPU call surface, header-only, on top of the C ABI (mifft.h).
//
// The reference is compiled code (Mojo); where its toolchain is absent the host side above the C ABI is C++ with the
// reference's names, argument meaning and error behaviour:
//
//   reference (fft/fft/fft.mojo)                                              here
//   plan_fft[in_dtype, out_dtype, in_layout, out_layout, *, bases, inverse]   mifftxx::plan_fft(in_dtype, out_dtype, in_layout,
//       (ctx: DeviceContext) raises -> _GPUPlan          (:161-210)               out_layout, ctx, bases, inverse) -> mifftxx::Plan
//   fft(output, x, ctx, *, plan) raises                  (:262-323)           mifftxx::fft(output, x, ctx, plan)
//   _estimate_best_bases[length, target]()               (:49-104)            mifftxx::estimate_best_bases(length, "gpu" | "cpu")
//   _get_ordered_bases_processed_list (fft/fft/_utils.mojo:186-221)           mifftxx::ordered_bases(length, bases)
//
// (Namespace mifftxx: `mifft` is the library's internal namespace.)
// Layouts are row-major (batches, d0[, d1 ...], C) with C_out = 2 and C_in in {1, 2} (fft/fft/fft.mojo:20-46); what the
// reference rejects with compile-time asserts (:22-46, _utils.mojo:189-220) throws mifftxx::Error with the reference's
// message BEFORE any device work.  `output` / `x` are device pointers on ctx.device; `fft` enqueues on ctx.stream and
// returns (the caller synchronises, fft/bench.mojo:51-52).  Link with -lmifft; nothing else is needed (no HIP headers).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mifft.h"

namespace mifftxx {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error("mifft error " + std::to_string(s) + ": " + m), status(s) {}
};

// DeviceContext of the reference: a device and the stream its work is enqueued on (NULL = the default stream)
struct DeviceContext {
    int device = 0;
    void* stream = nullptr;  // hipStream_t
};

inline void check(int rc) {
    if (rc < 0) throw Error(rc, mifft_last_error());
}

inline std::vector<uint32_t> ordered_bases(uint32_t length, const std::vector<uint32_t>& bases) {
    std::vector<uint32_t> out(MIFFT_MAX_STAGES);
    const int n = mifft_ordered_bases(length, bases.data(), (int)bases.size(), out.data(), (int)out.size());
    check(n);
    out.resize(n);
    return out;
}

inline std::vector<uint32_t> estimate_best_bases(uint32_t length, const std::string& target = "gpu") {
    std::vector<uint32_t> out(MIFFT_MAX_STAGES);
    const int n = mifft_estimate_bases(length, target == "gpu" ? 1 : 0, out.data(), (int)out.size());
    check(n);
    out.resize(n);
    return out;
}

// _GPUPlan (fft/fft/_ndim_fft_gpu.mojo:153-207): owns the device twiddle tables (and a scratch tensor on three routes only)
class Plan {
   public:
    Plan() = default;
    Plan(const Plan&) = delete;
    Plan& operator=(const Plan&) = delete;
    Plan(Plan&& o) noexcept { *this = std::move(o); }
    Plan& operator=(Plan&& o) noexcept {
        if (this != &o) {
            reset();
            h_ = o.h_;
            in_layout = std::move(o.in_layout);
            out_layout = std::move(o.out_layout);
            o.h_ = nullptr;
        }
        return *this;
    }
    ~Plan() { reset(); }

    std::vector<uint32_t> stages(int dim) const {
        std::vector<uint32_t> out(MIFFT_MAX_STAGES);
        const int n = mifft_plan_stages(h_, dim, out.data(), (int)out.size());
        check(n);
        out.resize(n);
        return out;
    }
    std::string kernel_name(int dim) const { return mifft_plan_kernel_name(h_, dim); }
    int num_launches() const { return mifft_plan_num_launches(h_); }
    size_t in_bytes() const { return mifft_plan_in_bytes(h_); }
    size_t out_bytes() const { return mifft_plan_out_bytes(h_); }
    size_t scratch_bytes() const { return mifft_plan_scratch_bytes(h_); }
    const mifft_plan* handle() const { return h_; }

    std::vector<int64_t> in_layout, out_layout;

   private:
    friend Plan plan_fft(mifft_dtype, mifft_dtype, const std::vector<int64_t>&, const std::vector<int64_t>&,
                         const DeviceContext&, const std::vector<std::vector<uint32_t>>*, bool, uint32_t);
    void reset() {
        if (h_) mifft_plan_destroy(h_);
        h_ = nullptr;
    }
    mifft_plan* h_ = nullptr;
};

// _check_layout_conditions_nd (fft/fft/fft.mojo:20-46), with the reference's assert texts
inline void check_layout_conditions_nd(const std::vector<int64_t>& in_layout, const std::vector<int64_t>& out_layout) {
    const size_t rank = out_layout.size();
    if (rank <= 2) throw Error(MIFFT_ERR_BAD_RANK, "The rank should be bigger than 2.");
    if (in_layout.size() != rank) throw Error(MIFFT_ERR_BAD_RANK, "in_layout and out_layout must have equal rank");
    if (in_layout[rank - 1] != 1 && in_layout[rank - 1] != 2)
        throw Error(MIFFT_ERR_BAD_COMPONENTS, "The last dimension of in_layout should be 1 or 2");
    if (out_layout[rank - 1] != 2) throw Error(MIFFT_ERR_BAD_COMPONENTS, "out_layout must have the last dimension equal to 2");
    for (size_t i = 0; i + 1 < rank; ++i)
        if (in_layout[i] != out_layout[i])
            throw Error(MIFFT_ERR_BAD_DIM, "out_layout and in_layout should have the same shape before the last dimension");
    for (size_t i = 1; i + 1 < rank; ++i)
        if (out_layout[i] == 1) throw Error(MIFFT_ERR_BAD_DIM, "no inner dimension should be of size 1");
}

// plan_fft, GPU overload (fft/fft/fft.mojo:161-210).  bases == nullptr: the reference's default
// (_estimate_best_bases_nd[.., "gpu"]).  `runtime_twfs` / `max_cluster_size` have no MI355X meaning; `_test` maps to
// flags = MIFFT_FLAG_FAITHFUL_STAGES (the user's stages run literally, one LDS pass each).
inline Plan plan_fft(mifft_dtype in_dtype, mifft_dtype out_dtype, const std::vector<int64_t>& in_layout,
                     const std::vector<int64_t>& out_layout, const DeviceContext& ctx,
                     const std::vector<std::vector<uint32_t>>* bases = nullptr, bool inverse = false,
                     uint32_t flags = MIFFT_FLAG_NONE) {
    check_layout_conditions_nd(in_layout, out_layout);
    const int ndim = (int)out_layout.size() - 2;
    if (bases && (int)bases->size() != ndim)
        throw Error(MIFFT_ERR_NO_BASES, "The bases list should have the same outer size as the amount of internal dimensions. "
                                        "e.g. (batches, dim_0, dim_1, dim_2, 2) -> len(bases) == 3");
    std::vector<uint32_t> flat;
    std::vector<int32_t> lens;
    if (bases)
        for (const auto& b : *bases) {
            lens.push_back((int32_t)b.size());
            flat.insert(flat.end(), b.begin(), b.end());
        }
    Plan p;
    p.in_layout = in_layout;
    p.out_layout = out_layout;
    check(mifft_plan_create(&p.h_, ctx.device, in_dtype, out_dtype, ndim, out_layout.data() + 1, out_layout[0],
                            (int)in_layout.back(), inverse ? 1 : 0, bases ? flat.data() : nullptr,
                            bases ? lens.data() : nullptr, flags));
    return p;
}

// fft, GPU overload (fft/fft/fft.mojo:262-323): out of place, x never written, every element of output written;
// asynchronous on ctx.stream
inline void fft(void* output, const void* x, const DeviceContext& ctx, const Plan& plan) {
    check(mifft_exec(plan.handle(), x, output, ctx.stream));
}

}  // namespace mifftxx
