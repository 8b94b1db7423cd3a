// Host-side check of HostRows::copy (csrc/host_rows.hpp): what calib_set_problem_views' staged upload gathers from a
// list of per-view arrays must be, byte for byte, the np.vstack of the views (src/calibrate.py:277-282) -- for any chunk
// of the stacked matrix: chunk borders inside rows and inside views, empty views, one-row views, the last partial
// chunk. Plain C++ (g++), run by tests/test_host_cpu.py; no GPU.
#include "../../camera-calibration_amd/csrc/host_rows.hpp"

#include <cstdio>
#include <random>
#include <vector>

int main() {
    std::mt19937_64 rng(7);
    int bad = 0, cases = 0;
    for (int width = 2; width <= 3; ++width)
        for (int trial = 0; trial < 40; ++trial) {
            const int nviews = 1 + (int)(rng() % 60);
            std::vector<int64_t> offs(nviews + 1, 0);
            std::vector<std::vector<double>> views(nviews);
            std::vector<const double*> ptrs(nviews, nullptr);
            std::vector<double> flat;
            for (int v = 0; v < nviews; ++v) {
                const int rows = (rng() % 5 == 0) ? 0 : 1 + (int)(rng() % 700);
                views[v].resize((size_t)rows * width);
                for (auto& x : views[v]) x = (double)(rng() % 1000003) * 1e-3;
                flat.insert(flat.end(), views[v].begin(), views[v].end());
                offs[v + 1] = offs[v] + rows;
                ptrs[v] = rows ? views[v].data() : nullptr;           // an empty view may carry a null pointer
            }
            calib::HostRows gathered, stacked;
            gathered.views = ptrs.data(); gathered.offs = offs.data(); gathered.nviews = nviews; gathered.width = width;
            stacked.flat = flat.data(); stacked.width = width;
            const size_t bytes = flat.size() * 8;
            for (size_t chunk : {(size_t)1, (size_t)7, (size_t)24, (size_t)1000, (size_t)4096, (size_t)1 << 20}) {
                std::vector<char> a(chunk), b(chunk);
                for (size_t off = 0; off < bytes; off += chunk) {
                    const size_t n = std::min(chunk, bytes - off);
                    std::fill(a.begin(), a.end(), (char)0x55);
                    std::fill(b.begin(), b.end(), (char)0x55);
                    gathered.copy(a.data(), off, n);
                    stacked.copy(b.data(), off, n);
                    if (std::memcmp(a.data(), b.data(), chunk) != 0) {
                        if (bad < 5) std::printf("width %d trial %d chunk %zu offset %zu: gathered bytes differ\n", width, trial, chunk, off);
                        ++bad;
                    }
                    ++cases;
                    if (chunk == 1 && off > 4000) break;             // byte-sized chunks: the head of the matrix is enough
                }
            }
        }
    calib::HostRows none;
    if (none.present()) ++bad;
    std::printf("HostRows::copy: %d chunk copies against the stacked matrix: %s\n", cases, bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
