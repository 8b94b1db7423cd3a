// HostRows: how the staged upload of calib_set_problem / calib_set_problem_views reads the caller's correspondences
// (host code only; included by calib_lm.hip and by tests/host_cpp/host_rows_check.cpp).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace calib {

// A (rows, width) float64 matrix on the host: one flat array, or one piece per view (what the reference's callers hold:
// a list of per-view arrays). The upload reads either; with pieces, getSensorPoints' vstack (src/calibrate.py:277-282)
// happens inside the staged copy -- nothing is stacked on the host first.
struct HostRows {
    const double* flat = nullptr;
    const double* const* views = nullptr;     // views[i] -> the rows of view i (C-contiguous)
    const int64_t* offs = nullptr;            // CSR row offsets of the views, nviews + 1 entries
    int64_t nviews = 0;
    int width = 0;
    bool present() const { return flat != nullptr || views != nullptr; }
    // bytes [off, off + n) of the stacked matrix -> buf
    void copy(char* buf, size_t off, size_t n) const {
        if (flat) { std::memcpy(buf, reinterpret_cast<const char*>(flat) + off, n); return; }
        const size_t rb = (size_t)width * 8;
        int64_t v = (std::upper_bound(offs, offs + nviews + 1, (int64_t)(off / rb)) - offs) - 1;
        while (n > 0 && v < nviews) {
            const size_t b0 = (size_t)offs[v] * rb, b1 = (size_t)offs[v + 1] * rb;
            if (off < b1) {
                const size_t take = std::min(n, b1 - off);
                std::memcpy(buf, reinterpret_cast<const char*>(views[v]) + (off - b0), take);
                buf += take; off += take; n -= take;
            }
            ++v;
        }
    }
};

}  // namespace calib
