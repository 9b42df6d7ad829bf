// planner.cpp -- host-only radix planning for libmifft.
//
// Reproduces the reference's planning RULES (not its code) so that a user's
// `bases` list means the same thing here as there:
//   * multiplicity of a base in N       fft/fft/_utils.mojo:125-152
//   * stage ordering (descending)       fft/fft/_utils.mojo:162-183
//   * validity checks                   fft/fft/_utils.mojo:186-221
//   * default bases                     fft/fft/fft.mojo:49-104
#include <algorithm>
#include <numeric>

#include "mifft_internal.h"

namespace mifft {

static bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }

// How many copies of `base` the reference schedules for a length.
// Power-of-two bases use trailing-zero arithmetic (so 4 "divides" 8 once and
// 32 zero times... but 8 = 2^3 holds floor(3/2) = 1 fours); other bases count
// exact repeated divisibility, stopping at the first non-multiple -- with the
// reference's quirk that a quotient equal to the base counts once more and a
// quotient smaller than the base ends the count.
static uint64_t multiplicity(uint64_t length, uint64_t base) {
    if (is_pow2(base)) {
        if (base == 1 || length == 0) return 0;
        return (uint64_t)__builtin_ctzll(length) / (uint64_t)__builtin_ctzll(base);
    }
    uint64_t count = 0, x = length;
    for (;;) {
        if (x == base) return count + 1;
        if (base > x || x % base) return count;
        x /= base;
        ++count;
    }
}

int plan_ordered_bases(uint64_t length, const std::vector<uint64_t>& user, std::vector<uint32_t>& ordered,
                       std::vector<uint32_t>& processed, std::string& err) {
    ordered.clear();
    processed.clear();
    if (user.empty()) {
        err = "The amount of bases is not enough: []";
        return MIFFT_ERR_NO_BASES;
    }
    if (std::any_of(user.begin(), user.end(), [](uint64_t b) { return b < 2; })) {
        err = "Cannot do an fft with base 1.";
        return MIFFT_ERR_BASE_ONE;
    }
    std::vector<uint64_t> desc(user);
    std::sort(desc.begin(), desc.end(), std::greater<uint64_t>());
    uint64_t product = std::accumulate(desc.begin(), desc.end(), (uint64_t)1, std::multiplies<uint64_t>());
    std::vector<uint64_t> stages;
    if (product == length) {
        stages = desc;  // the list already factors N: use it, largest radix first
    } else {
        uint64_t done = 1;
        for (uint64_t b : desc) {
            uint64_t m = multiplicity(length, b);  // NB: of the full length, like the reference
            stages.insert(stages.end(), m, b);
            for (uint64_t k = 0; k < m; ++k) done *= b;
            if (done == length) break;
        }
    }
    uint64_t p = 1;
    for (uint64_t r : stages) {
        ordered.push_back((uint32_t)r);
        processed.push_back((uint32_t)p);
        p *= r;
    }
    if (stages.empty() || p != length) {
        err = "powers of the bases must multiply together to equal the sequence length. "
              "The builtin algorithm was only able to produce: [";
        for (size_t i = 0; i < ordered.size(); ++i) err += (i ? ", " : "") + std::to_string(ordered[i]);
        err += "] for the length: " + std::to_string(length);
        ordered.clear();
        processed.clear();
        return MIFFT_ERR_BAD_BASES;
    }
    if ((int)ordered.size() > MIFFT_MAX_STAGES) {
        err = "too many stages";
        return MIFFT_ERR_BAD_BASES;
    }
    return MIFFT_OK;
}

// Greedy factor collection over a candidate list, dividing the remaining length
// as it goes; returns the collected factors in reverse collection order (the
// reference reverses its list before returning).
static bool collect(uint64_t length, const std::vector<uint64_t>& candidates, std::vector<uint64_t>& out) {
    out.clear();
    uint64_t done = 1;
    for (uint64_t r : candidates) {
        uint64_t m = multiplicity(length / done, r);
        out.insert(out.end(), m, r);
        for (uint64_t k = 0; k < m; ++k) done *= r;
        if (done == length) {
            std::reverse(out.begin(), out.end());
            return true;
        }
    }
    return false;
}

std::vector<uint64_t> plan_estimate_bases(uint64_t length, bool gpu_target) {
    std::vector<uint64_t> out;
    if (gpu_target && length / 32 <= 1024) {
        // radices ceil(N/1024) (at least 2) .. 32, ascending trial
        std::vector<uint64_t> cand;
        for (uint64_t r = std::max<uint64_t>((length + 1023) / 1024, 2); r <= 32; ++r) cand.push_back(r);
        if (collect(length, cand, out)) return out;
    }
    static const uint64_t primes_desc[] = {97, 89, 83, 79, 73, 71, 67, 61, 59, 53, 47, 43, 41,
                                           37, 31, 29, 23, 19, 17, 13, 11, 7,  5,  3,  2};
    std::vector<uint64_t> cand(std::begin(primes_desc), std::end(primes_desc));
    collect(length, cand, out);  // incomplete when N has a prime factor > 97: caller's check rejects it
    return out;
}

}  // namespace mifft
