// C-ABI host side of the LM refinement engine (see include/calib_lm.h).
// Owns device memory, packs correspondences to SoA, sequences the kernels of kernels.hpp.
#include "../../include/calib_lm.h"
#include "kernels.hpp"
#include "host_rows.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <chrono>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

using namespace calib;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(CALIB_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

// after a kernel launch: a launch-time failure is reported with the kernel's name, and the name is remembered so
// that an ASYNCHRONOUS fault -- seen only at a later synchronisation -- can say what was in flight
#define LAUNCHED(h, name)                                                                     \
    do {                                                                                      \
        if (h) (h)->noteKernel(name);                                                         \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess)                                                                \
            return fail(CALIB_E_HIP, std::string("launch of ") + (name) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define SYNC_H(h)                                                                             \
    do {                                                                                      \
        hipError_t e__ = hipStreamSynchronize((h)->stream);                                   \
        if (e__ != hipSuccess)                                                                \
            return fail(CALIB_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e__) + \
                                     " (kernels enqueued since the last successful synchronisation: " + (h)->recentKernels() + ")"); \
        (h)->kernels_since_sync.clear();                                                      \
    } while (0)

#define CHECK_H(h)                                                                            \
    if (!(h)) return fail(CALIB_E_INVALID, "null handle");                                    \
    HIP_TRY(hipSetDevice((h)->device))

// Device allocation owned by its scope: every early return of an entry point (HIP_TRY) releases it.
template <typename U>
struct DevBuf {
    U* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(U));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

constexpr int kEventPool = 32768;

}  // namespace

struct calib_handle_s {
    // names of the kernels enqueued since the last successful synchronisation (distinct, in first-use order)
    std::vector<const char*> kernels_since_sync;
    void noteKernel(const char* name) {
        for (const char* k : kernels_since_sync) if (k == name) return;
        if (kernels_since_sync.size() < 16) kernels_since_sync.push_back(name);
    }
    std::string recentKernels() const {
        std::string r;
        for (const char* k : kernels_since_sync) { if (!r.empty()) r += ", "; r += k; }
        return r.empty() ? "none" : r;
    }
    int model = 0, dtype = 0, device = 0;
    int L = 10, C = 16;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // problem
    bool has_problem = false;
    int64_t M = 0, MN = 0;        // external views, points
    int nv = 0;                   // non-empty views
    int n_items = 0;
    int64_t n_tiles = 0;
    int max_views_per_tile = 1;
    int schur_blocks = 1;
    int gram_wpi = 1;             // waves per gram item (two-kernel mode)
    int fused_wpi = 1;            // waves per item of the fused kernel
    int uniform_n = 0;            // > 0: every item is one whole view of exactly this many points, in order (item i = view i = points [i n, (i+1) n))
    int head_loads = 0;           // per-view kernels' record-head loads: 0 = by shard size, 1 = one load per value, 2 = coalesced + DPP (CALIB_HEAD_LOADS)
    int items_per_wave = 0;       // fused kernel, short uniform items: 0 = chosen per shard (CALIB_ITEMS_PER_WAVE)
    int upd_lane_views = 0;       // shards from this many views on take update_backsub_lane_kernel (CALIB_UPD_LANE_VIEWS)
    int upd_small_views = 0;      // shards up to this many views take update_backsub_small_kernel (CALIB_UPD_SMALL_VIEWS)
    int gram_form = 0;            // fp64 fused kernel: 0 = chosen per shard, 1 = 16x16x4 tiles, 2 = 4x4x4 blocks (CALIB_GRAM_FORM)
    int stream_mode = -1;         // fused_stream_kernel: -1 = chosen per shard, 0 = never, 1 = whenever the shard allows it (CALIB_FUSED_STREAM)
    int stream_waves_env = 0;     // > 0: waves of the stream launch (CALIB_STREAM_WAVES); 0 = the chip's wave slots
    int stream_share = 0;         // > 0: this problem's fused rounds run in stream form, `stream_share` 4-point groups per wave
    int stream_waves = 0;         // waves that have work = overflow records behind the nv view records
    // LM rounds walk the points in chunks of whole views so that a chunk's compact J
    // (written by the jacobian kernel, read once by the gram kernel) can stay on-die
    struct Chunk { int64_t p0, p1; int item0, item1; };
    std::vector<Chunk> chunks;
    int64_t max_chunk_points = 0;
    DevBuf<unsigned char> uv, XY, Z, VC, J, r, y;   // typed by dtype
    DevBuf<int> pt_view, view_ext, item_n, view_item0, item_view;
    DevBuf<uint32_t> emit_tab;    // fused kernel's record assembly table (buildEmitTable)
    DevBuf<int32_t> stream_ops;   // fused_stream_kernel's per-lane record offsets (buildStreamOps)
    int lm_mode = CALIB_LM_FUSED;
    int num_cus = 256;
    DevBuf<int64_t> item_pt0;
    DevBuf<double> sse_part, G[2], bpart, part, red_own, P[2], Peval, trace;
    int n_bpart = 0;              // workgroup partials of the shared block written by this round's pass
    DevBuf<LMState> st, st_eval;
    double* red = nullptr;        // active reduce buffer (own or bound)
    void* comm = nullptr;         // RCCL communicator of the in-library all-reduce (calib_rccl_init), or null
    DevBuf<double> rccl_test;     // operand of calib_rccl_selftest: outlives a collective that timed out
    int comm_ranks = 0, comm_rank = 0;
    // peer exchange over xGMI (calib_peer_*): the reduce kernel sums over the ranks itself
    void* peer_mem = nullptr;             // this rank's slot memory (uncached / fine-grained device memory)
    std::vector<void*> peer_open;         // IPC mappings of the other ranks' slot memory
    DevBuf<unsigned long long*> peer_slots;
    std::vector<unsigned long long*> peer_slot_host;     // the same pointers, for the kernel arguments
    DevBuf<int> peer_flags;               // [0] a spin timed out, [1] self-test mismatches
    int peer_world = 0, peer_rank = 0;
    unsigned peer_epoch = 0;
    double peer_timeout_s = 60.0;
    bool peer_connected = false;
    bool exchange_round = false;          // the round being enqueued belongs to a sharded run
    int* host_done = nullptr;             // pinned, device-visible: 1 = the update kernel says the LM loop is over, 2 = a peer exchange gave up
    int* host_done_dev = nullptr;         // the same word as the device sees it
    bool lm_active = false;
    int lm_max_iters = 0;
    int rounds_enqueued = 0;

    // pinned staging of calib_set_problem's uploads (upload_staged)
    bool stage_ready = false;
    void* stage_pinned = nullptr;
    hipStream_t stage_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    void* stage_buf[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    hipEvent_t stage_ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};

    // profiling
    bool prof = false;
    int prof_stride = 1;          // every prof_stride-th launch of a kernel kind is timed
    int64_t prof_seen[3] = {0, 0, 0};
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> ev_kind;
    size_t ev_used = 0;
};

namespace {

size_t tsize(const calib_handle_s* h) { return h->dtype == CALIB_DTYPE_F64 ? 8 : 4; }

// Profiled launches (calib_profile_enable): every prof_stride-th launch of a kernel kind gets a HIP event pair that
// rides on the kernel's own dispatch (hipExtLaunchKernelGGL: the events take the kernel's begin and end timestamps).
// Recording stream markers around the launch instead put two more packets into the queue, kept the launch from
// being dispatched back to back with its neighbours and read ~3 us long on a 45 us kernel.
int prof_reserve(calib_handle_s* h, int kind) {
    if (!h->prof || h->ev_used + 2 > h->ev.size()) return -1;
    if (h->prof_seen[kind]++ % h->prof_stride != 0) return -1;
    int idx = (int)h->ev_used;
    h->ev_used += 2;
    h->ev_kind[idx / 2] = kind;
    return idx;
}

template <typename F, typename... Args>
void launch_kind(calib_handle_s* h, int kind, F kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    const int idx = prof_reserve(h, kind);
    if (idx >= 0)
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, h->stream, h->ev[(size_t)idx], h->ev[(size_t)idx + 1], 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, shmem, h->stream, args...);
}

// ---- RCCL, resolved at run time (calib_rccl_load) ------------------------------------------
// Only the handful of entry points the one all-reduce needs; types as in rccl.h (ncclUniqueId is 128
// opaque bytes passed by value, ncclDouble = 8, ncclSum = 0, ncclSuccess = 0).
struct RcclId { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*getUniqueId)(RcclId*) = nullptr;
    int (*commInitRank)(void**, int, RcclId, int) = nullptr;
    int (*allReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*commDestroy)(void*) = nullptr;
    int (*commAbort)(void*) = nullptr;
    const char* (*getErrorString)(int) = nullptr;
};
RcclApi g_rccl;                 // written once under g_rccl_mutex (calib_rccl_load), read-only afterwards
std::mutex g_rccl_mutex;
constexpr int kNcclDouble = 8, kNcclSum = 0;

int rccl_fail(const char* what, int rc) {
    return fail(CALIB_E_HIP, std::string(what) + ": " +
                             (g_rccl.getErrorString ? g_rccl.getErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

// ---- host -> device copies of large caller arrays --------------------------------------------
// Pageable memory goes over PCIe at ~8 GB/s through hipMemcpy. Here kUploadThreads host threads copy
// alternating 1 MiB chunks into their own pair of pinned buffers and DMA them from there on the handle's stream:
// the host memcpy of one chunk overlaps the DMA of the previous ones (measured: 31-40 GB/s). The pinned memory
// is one 8 MiB allocation per handle -- pinning costs ~0.3 ms per MiB, paid by the handle's first upload.
constexpr int kUploadThreads = 4;
constexpr size_t kUploadChunk = (size_t)1 << 20;

int upload_staged(calib_handle_s* h, void* dst, const HostRows& src, size_t bytes) {
    if (bytes == 0) return CALIB_OK;
    if (bytes < 2 * kUploadChunk) {
        if (src.flat) {
            HIP_TRY(hipMemcpy(dst, src.flat, bytes, hipMemcpyHostToDevice));
        } else {
            std::vector<char> tmp(bytes);
            src.copy(tmp.data(), 0, bytes);
            HIP_TRY(hipMemcpy(dst, tmp.data(), bytes, hipMemcpyHostToDevice));
        }
        return CALIB_OK;
    }
    if (!h->stage_ready) {
        HIP_TRY(hipHostMalloc(&h->stage_pinned, 2 * kUploadThreads * kUploadChunk, hipHostMallocDefault));
        for (int t = 0; t < kUploadThreads; ++t) {
            h->stage_stream[t] = h->own_stream;       // one DMA queue saturates the link; creating streams costs ms each
            for (int b = 0; b < 2; ++b) {
                h->stage_buf[t][b] = static_cast<char*>(h->stage_pinned) + (size_t)(2 * t + b) * kUploadChunk;
                HIP_TRY(hipEventCreateWithFlags(&h->stage_ev[t][b], hipEventDisableTiming));
            }
        }
        h->stage_ready = true;
    }
    const size_t nchunks = (bytes + kUploadChunk - 1) / kUploadChunk;
    hipError_t errs[kUploadThreads];
    std::thread workers[kUploadThreads];
    for (int t = 0; t < kUploadThreads; ++t) {
        errs[t] = hipSuccess;
        workers[t] = std::thread([=, &errs, &src]() {
            hipError_t e = hipSetDevice(h->device);
            size_t use = 0;
            for (size_t c = (size_t)t; c < nchunks && e == hipSuccess; c += kUploadThreads, ++use) {
                const int b = (int)(use & 1);
                const size_t off = c * kUploadChunk, n = std::min(kUploadChunk, bytes - off);
                if (use >= 2) e = hipEventSynchronize(h->stage_ev[t][b]);      // this buffer's previous DMA is done
                if (e != hipSuccess) break;
                src.copy(static_cast<char*>(h->stage_buf[t][b]), off, n);
                e = hipMemcpyAsync(static_cast<char*>(dst) + off, h->stage_buf[t][b], n, hipMemcpyHostToDevice,
                                   h->stage_stream[t]);
                if (e == hipSuccess) e = hipEventRecord(h->stage_ev[t][b], h->stage_stream[t]);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(h->stage_stream[t]);
            errs[t] = e;
        });
    }
    for (auto& w : workers) w.join();
    for (int t = 0; t < kUploadThreads; ++t)
        if (errs[t] != hipSuccess) return fail(CALIB_E_HIP, std::string("staged upload: ") + hipGetErrorString(errs[t]));
    return CALIB_OK;
}

// ---- launches ---------------------------------------------------------------------------
template <typename T>
int launch_view_setup(calib_handle_s* h, const double* P0, const double* P1, const LMState* st, int sel) {
    if (h->nv == 0) return CALIB_OK;
    const int threads = 64;
    const int blocks = (h->nv + threads - 1) / threads;
    hipLaunchKernelGGL((view_setup_kernel<T>), dim3(blocks), dim3(threads), 0, h->stream, P0, P1, st, sel,
                       h->L, h->view_ext.p, h->nv, reinterpret_cast<T*>(h->VC.p));
    LAUNCHED(h, "view_setup_kernel");
    return CALIB_OK;
}

template <int MODEL, typename T>
int launch_jacobian_t(calib_handle_s* h, const double* P0, const double* P1, const LMState* st, int sel,
                      bool wantJ, bool wantR, bool wantY, bool wantSse, int64_t p_begin, int64_t p_end) {
    using T2 = typename Pair<T>::type;
    if (p_end <= p_begin) return CALIB_OK;
    JacArgs<T> a;
    a.P0 = P0; a.P1 = P1; a.st = st; a.sel = sel;
    a.uv = reinterpret_cast<const T2*>(h->uv.p);
    a.XY = reinterpret_cast<const T2*>(h->XY.p);
    a.Z = reinterpret_cast<const T*>(h->Z.p);
    a.pt_view = h->pt_view.p;
    a.p_begin = p_begin;
    a.p_end = p_end;
    a.VC = reinterpret_cast<const T*>(h->VC.p);
    a.J = wantJ ? reinterpret_cast<T2*>(h->J.p) : nullptr;
    a.r = wantR ? reinterpret_cast<T2*>(h->r.p) : nullptr;
    a.y = wantY ? reinterpret_cast<T2*>(h->y.p) : nullptr;
    a.sse_part = wantSse ? h->sse_part.p : nullptr;
    const size_t lds = 32 + (size_t)h->max_views_per_tile * kViewStride * sizeof(T);
    const unsigned tiles = (unsigned)((p_end - p_begin + kTile - 1) / kTile);
    launch_kind(h, 0, jacobian_kernel<MODEL, T>, dim3(tiles), dim3(kTile), lds, a);
    LAUNCHED(h, "jacobian_kernel");
    return CALIB_OK;
}

int launch_jacobian(calib_handle_s* h, const double* P0, const double* P1, const LMState* st, int sel,
                    bool wantJ, bool wantR, bool wantY, bool wantSse, int64_t p_begin, int64_t p_end) {
    if (h->dtype == CALIB_DTYPE_F64) {
        return h->model == CALIB_MODEL_RADTAN
                   ? launch_jacobian_t<kRadtan, double>(h, P0, P1, st, sel, wantJ, wantR, wantY, wantSse, p_begin, p_end)
                   : launch_jacobian_t<kFisheye, double>(h, P0, P1, st, sel, wantJ, wantR, wantY, wantSse, p_begin, p_end);
    }
    return h->model == CALIB_MODEL_RADTAN
               ? launch_jacobian_t<kRadtan, float>(h, P0, P1, st, sel, wantJ, wantR, wantY, wantSse, p_begin, p_end)
               : launch_jacobian_t<kFisheye, float>(h, P0, P1, st, sel, wantJ, wantR, wantY, wantSse, p_begin, p_end);
}

int launch_view_setup_any(calib_handle_s* h, const double* P0, const double* P1, const LMState* st, int sel) {
    return h->dtype == CALIB_DTYPE_F64 ? launch_view_setup<double>(h, P0, P1, st, sel)
                                       : launch_view_setup<float>(h, P0, P1, st, sel);
}

template <typename T, int C>
int launch_gram_t(calib_handle_s* h, const LMState* st, int sel, int item0, int item1, int64_t origin) {
    using T2 = typename Pair<T>::type;
    if (item1 <= item0) return CALIB_OK;
    const int ipb = 4 / h->gram_wpi;       // items per workgroup
    const int blocks = (item1 - item0 + ipb - 1) / ipb;
    launch_kind(h, 1, gram_kernel<T, C>, dim3(blocks), dim3(256), 0,
                reinterpret_cast<const T2*>(h->J.p), reinterpret_cast<const T2*>(h->r.p),
                (const int64_t*)h->item_pt0.p, (const int*)h->item_n.p, item0, item1, origin, h->gram_wpi, st, sel, h->G[0].p,
                h->G[1].p, h->bpart.p, h->n_bpart);
    h->n_bpart += blocks;                  // the chunk's workgroups append their partials
    LAUNCHED(h, "gram_kernel");
    return CALIB_OK;
}

int launch_gram(calib_handle_s* h, const LMState* st, int sel, int item0, int item1, int64_t origin) {
    if (h->dtype == CALIB_DTYPE_F64)
        return h->model == CALIB_MODEL_RADTAN ? launch_gram_t<double, 16>(h, st, sel, item0, item1, origin)
                                              : launch_gram_t<double, 15>(h, st, sel, item0, item1, origin);
    return h->model == CALIB_MODEL_RADTAN ? launch_gram_t<float, 16>(h, st, sel, item0, item1, origin)
                                          : launch_gram_t<float, 15>(h, st, sel, item0, item1, origin);
}

template <int MODEL, typename T>
int launch_fused_t(calib_handle_s* h, const LMState* st, int sel) {
    using T2 = typename Pair<T>::type;
    if (h->n_items == 0) return CALIB_OK;
    // ROWS = 32, 4 waves per workgroup: the 64-row / 2-wave variants measured 1-6 % slower (c3, c5, c2)
    const int wpi = std::min(h->fused_wpi, 4);
    const bool g44 = sizeof(T) == 8 && (h->gram_form == 2 || (h->gram_form == 0 && h->MN > (int64_t)128 * h->n_items));
    // short uniform items (tile forms, one wave each): on shards large enough to leave every workgroup slot of the
    // chip (4 per CU) four workgroups even so, a wave takes up to four items in a row -- one prologue, one partial,
    // one barrier, the next item's points requested early (c5 shard -7 %; c4's 12 500 items: no gain, c2: slower)
    int ipw = 1;
    if (!g44 && wpi == 1 && h->uniform_n > 0) {
        if (h->items_per_wave > 0) ipw = h->items_per_wave;
        else while (ipw < 4 && h->n_items / (8 * ipw) >= 16 * h->num_cus) ipw *= 2;
    }
    const int ipb = (4 / wpi) * ipw;
    const int blocks = (h->n_items + ipb - 1) / ipb;
    // fp64 items of more than two batches build J^T J from 4x4 blocks (v_mfma_f64_4x4x4_4b, symmetric half only;
    // c3 -4.5 %); shorter items stay on the 16x16x4 form, whose record goes to HBM straight from the accumulators
    // (one-batch items: c2 +4 % on the block form; two batches, c5: no difference)
    auto launch = [&](auto kernel) {
        launch_kind(h, 2, kernel, dim3(blocks), dim3(256), 0, (const double*)h->P[0].p,
                    (const double*)h->P[1].p, reinterpret_cast<const T2*>(h->uv.p), reinterpret_cast<const T2*>(h->XY.p),
                    reinterpret_cast<const T*>(h->Z.p), reinterpret_cast<const T*>(h->VC.p), (const int64_t*)h->item_pt0.p,
                    (const int*)h->item_n.p, (const int*)h->item_view.p, h->n_items, h->uniform_n, ipw, wpi,
                    (const uint32_t*)h->emit_tab.p, st, sel, h->G[0].p, h->G[1].p, h->bpart.p);
    };
    if constexpr (sizeof(T) == 8) {
        if (g44) launch(fused_kernel<MODEL, T, 32, 4, true, false>);
        else if (ipw > 1) launch(fused_kernel<MODEL, T, 32, 4, false, true>);
        else launch(fused_kernel<MODEL, T, 32, 4, false, false>);
    } else {
        if (ipw > 1) launch(fused_kernel<MODEL, T, 32, 4, false, true>);
        else launch(fused_kernel<MODEL, T, 32, 4, false, false>);
    }
    h->n_bpart = blocks;
    LAUNCHED(h, "fused_kernel");
    return CALIB_OK;
}

// records of this problem's fused rounds: stream form (view records + one overflow record per wave) or one per item
bool stream_rounds(const calib_handle_s* h) { return h->lm_mode == CALIB_LM_FUSED && h->stream_share > 0; }
StreamMap stream_map(const calib_handle_s* h) {
    StreamMap sm;
    sm.share = stream_rounds(h) ? h->stream_share : 0;
    sm.n4 = h->uniform_n / 4;
    sm.nv = h->nv;
    return sm;
}
int num_records(const calib_handle_s* h) { return std::max(h->n_items, 1) + h->stream_waves; }

template <int MODEL>
int launch_fused_stream(calib_handle_s* h, const LMState* st, int sel) {
    const int blocks = (h->stream_waves + 3) / 4;
    launch_kind(h, 2, fused_stream_kernel<MODEL>, dim3(blocks), dim3(256), 0, (const double*)h->P[0].p, (const double*)h->P[1].p,
                reinterpret_cast<const double2*>(h->uv.p), reinterpret_cast<const double2*>(h->XY.p),
                reinterpret_cast<const double*>(h->Z.p), reinterpret_cast<const double*>(h->VC.p), h->uniform_n,
                h->nv, h->stream_share, (const uint32_t*)h->emit_tab.p, (const int32_t*)h->stream_ops.p, st, sel, h->G[0].p, h->G[1].p,
                h->bpart.p);
    h->n_bpart = blocks;
    LAUNCHED(h, "fused_stream_kernel");
    return CALIB_OK;
}

int launch_fused(calib_handle_s* h, const LMState* st, int sel) {
    if (stream_rounds(h))
        return h->model == CALIB_MODEL_RADTAN ? launch_fused_stream<kRadtan>(h, st, sel)
                                              : launch_fused_stream<kFisheye>(h, st, sel);
    if (h->dtype == CALIB_DTYPE_F64)
        return h->model == CALIB_MODEL_RADTAN ? launch_fused_t<kRadtan, double>(h, st, sel)
                                              : launch_fused_t<kFisheye, double>(h, st, sel);
    return h->model == CALIB_MODEL_RADTAN ? launch_fused_t<kRadtan, float>(h, st, sel)
                                          : launch_fused_t<kFisheye, float>(h, st, sel);
}

// per-view kernels skip the view -> item indirection when every view is a single item
const int* view_items(const calib_handle_s* h) { return h->n_items == h->nv ? nullptr : h->view_item0.p; }

// arguments of the next peer exchange: every rank enqueues the same sequence of exchanges, so the epoch
// counters advance in lockstep
PeerExchange next_exchange(calib_handle_s* h) {
    PeerExchange x;
    x.slots = h->peer_slots.p;
    for (int r = 0; r < kPeerInline; ++r) x.slot8[r] = r < (int)h->peer_slot_host.size() ? h->peer_slot_host[(size_t)r] : nullptr;
    x.fault = h->peer_flags.p;
    x.notify = h->host_done_dev;
    x.timeout_ticks = (unsigned long long)(h->peer_timeout_s * 1e8);      // wall_clock64 counts at 100 MHz
    x.epoch = ++h->peer_epoch;
    x.world = h->peer_world;
    x.rank = h->peer_rank;
    return x;
}

// shards above this many views load record heads in the coalesced, DPP-broadcast form (kernels.hpp: load_view_head).
// Round 4: since a record is six rows (768 B) the broadcast chain behind the loads is 48 DPP moves instead of 54, and on
// c4's 12 500-view shard the wide update kernel (122 VGPRs, no scratch) is as fast as the narrow one was WITH its 12-byte
// spill (10.1 vs 10.3 us; the narrow one without the spill, at three workgroups per CU: 12.3 us -- 782 workgroups on 768
// slots are two rounds). So every shard the small update kernel does not take (> 4 096 views) loads wide; the narrow
// forms remain for the small shards and on request (CALIB_HEAD_LOADS=narrow).
constexpr int kWideHeadViews = 4096;

// Larger shards, and shards whose views can be two records (stream form: the second record doubles the 27 per-value
// loads of the narrow form; c3 schur +2.1 us, update +3 us -- six coalesced rows per record cost nothing extra)
bool wide_heads(const calib_handle_s* h) {
    return h->head_loads == 2 || (h->head_loads == 0 && (h->nv > kWideHeadViews || stream_rounds(h)));
}

int launch_schur(calib_handle_s* h, const LMState* st) {
    dim3 grid(h->schur_blocks, 3);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(kSchurBlock), 0, h->stream, h->G[0].p, h->G[1].p, st, view_items(h), h->nv,
                           stream_map(h), h->bpart.p, h->n_bpart, h->part.p);
    };
    const bool wide = wide_heads(h), strm = stream_rounds(h);
    auto pick = [&](auto Lc) {
        constexpr int LL = decltype(Lc)::value;
        if (strm) { if (wide) launch(schur_kernel<LL, true, true>); else launch(schur_kernel<LL, false, true>); }
        else { if (wide) launch(schur_kernel<LL, true, false>); else launch(schur_kernel<LL, false, false>); }
    };
    if (h->L == 10) pick(std::integral_constant<int, 10>{}); else pick(std::integral_constant<int, 9>{});
    LAUNCHED(h, "schur_kernel");
    return CALIB_OK;
}

int launch_schur_reduce(calib_handle_s* h, const LMState* st, double* red) {
    const int VA = variantSize(h->L);
    if (h->nv > 0) {
        const int rc = launch_schur(h, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(reduce_kernel, dim3(2 * VA), dim3(64), 0, h->stream, h->part.p,
                       h->nv > 0 ? h->schur_blocks : 0, VA, st, red,
                       h->exchange_round ? next_exchange(h) : PeerExchange{});
    LAUNCHED(h, "reduce_kernel");
    return CALIB_OK;
}

// the LM state is double-buffered per round: kernels of round r read st[r & 1], the update kernel
// writes st[(r + 1) & 1]
LMState* st_cur(calib_handle_s* h) { return h->st.p + (h->rounds_enqueued & 1); }
LMState* st_next(calib_handle_s* h) { return h->st.p + ((h->rounds_enqueued + 1) & 1); }

// shards of at most this many views take the latency-oriented form of the update kernel (kernels.hpp)
constexpr int kUpdSmallViews = 4096;
// shards of at least this many views take the one-lane-per-view form (a wave per 64 views: it needs many to fill the chip).
// tools/sweep_upd_lane.sh, update kernel us, 16 lanes per view / one lane per view: 10 000 views 8.9 / 15.0 - 12 500 (fp32)
// 10.0 / 11.5 - 16 384: 11.2 / 12.4 - 32 768: 19.0 / 13.3 - 65 536: 29.7 / 17.6 - 125 000: 47.5 / 30.5
constexpr int kUpdLaneViews = 24576;

template <int L, typename T>
int launch_update_backsub_t(calib_handle_s* h) {
    if (h->nv <= h->upd_small_views) {
        const int blocks = std::max(1, (h->nv + kUpdViewsPerBlock - 1) / kUpdViewsPerBlock);    // one view per 16-lane group
        auto launchSmall = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kUpdThreads), 0, h->stream,
                               h->G[0].p, h->G[1].p, st_cur(h), st_next(h), h->red, view_items(h), h->view_ext.p, h->nv,
                               stream_map(h), h->P[0].p, h->P[1].p, h->trace.p, reinterpret_cast<T*>(h->VC.p));
        };
        if (stream_rounds(h)) launchSmall(update_backsub_small_kernel<L, T, true>);
        else launchSmall(update_backsub_small_kernel<L, T, false>);
        LAUNCHED(h, "update_backsub_small_kernel");
        return CALIB_OK;
    }
    if (h->nv >= h->upd_lane_views && h->head_loads == 0) {
        const int blocks = std::max(1, std::min(12 * h->num_cus, (h->nv + kSchurThreads - 1) / kSchurThreads));
        auto launchLane = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kSchurThreads), 0, h->stream,
                               h->G[0].p, h->G[1].p, st_cur(h), st_next(h), h->red, view_items(h), h->view_ext.p, h->nv,
                               stream_map(h), h->P[0].p, h->P[1].p, h->trace.p, reinterpret_cast<T*>(h->VC.p));
        };
        if (stream_rounds(h)) launchLane(update_backsub_lane_kernel<L, T, true>);
        else launchLane(update_backsub_lane_kernel<L, T, false>);
        LAUNCHED(h, "update_backsub_lane_kernel");
        return CALIB_OK;
    }
    const int per = kSchurThreads / 16;
    const int blocks = std::max(1, std::min(2048, (h->nv + per - 1) / per));     // grid-stride over views
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kSchurThreads), 0, h->stream, h->G[0].p,
                           h->G[1].p, st_cur(h), st_next(h), h->red, view_items(h), h->view_ext.p, h->nv,
                           stream_map(h), h->P[0].p, h->P[1].p, h->trace.p, reinterpret_cast<T*>(h->VC.p));
    };
    if (stream_rounds(h)) { if (wide_heads(h)) launch(update_backsub_kernel<L, T, true, true>); else launch(update_backsub_kernel<L, T, false, true>); }
    else { if (wide_heads(h)) launch(update_backsub_kernel<L, T, true, false>); else launch(update_backsub_kernel<L, T, false, false>); }
    LAUNCHED(h, "update_backsub_kernel");
    return CALIB_OK;
}

int launch_update_backsub(calib_handle_s* h) {
    if (h->dtype == CALIB_DTYPE_F64)
        return h->L == 10 ? launch_update_backsub_t<10, double>(h) : launch_update_backsub_t<9, double>(h);
    return h->L == 10 ? launch_update_backsub_t<10, float>(h) : launch_update_backsub_t<9, float>(h);
}

int need_problem(calib_handle_s* h) {
    if (!h->has_problem) return fail(CALIB_E_STATE, "calib_set_problem has not been called");
    return CALIB_OK;
}

int64_t numParams(const calib_handle_s* h) { return h->L + 6 * h->M; }

}  // namespace

// ============================================================================ C-ABI
extern "C" {

int calib_version(void) { return 400; }   // 4.0: 768-byte records, calib_set_problem_views, one-lane-per-view update kernel

const char* calib_last_error(void) { return g_err.c_str(); }

int calib_device_count(int* out_count) {
    if (!out_count) return fail(CALIB_E_INVALID, "out_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fail(CALIB_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *out_count = n;
    return CALIB_OK;
}

int calib_create(int model, int dtype, int device_id, calib_handle_t* out_handle) {
    if (!out_handle) return fail(CALIB_E_INVALID, "out_handle is null");
    *out_handle = nullptr;
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE)
        return fail(CALIB_E_INVALID, "unknown distortion model");
    if (dtype != CALIB_DTYPE_F64 && dtype != CALIB_DTYPE_F32)
        return fail(CALIB_E_INVALID, "unknown dtype");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(CALIB_E_HIP, "no such HIP device (this library has no CPU fallback)");
    HIP_TRY(hipSetDevice(device_id));
    calib_handle_s* h = new (std::nothrow) calib_handle_s();
    if (!h) return fail(CALIB_E_INVALID, "out of host memory");
    h->model = model;
    h->dtype = dtype;
    h->device = device_id;
    h->L = model == CALIB_MODEL_RADTAN ? 10 : 9;
    h->C = h->L + 6;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(CALIB_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    h->stream = h->own_stream;
    {
        uint32_t tab[kEmitTabSize];
        buildEmitTable(h->C, tab);
        e = h->emit_tab.alloc(kEmitTabSize);
        if (e == hipSuccess) e = hipMemcpy(h->emit_tab.p, tab, sizeof(tab), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            h->emit_tab.release();
            (void)hipStreamDestroy(h->own_stream);
            delete h;
            return fail(CALIB_E_HIP, std::string("emit table: ") + hipGetErrorString(e));
        }
    }
    {
        int32_t ops[64 * kStreamOps];
        const bool built = buildStreamOps(h->C, ops);
        e = built ? h->stream_ops.alloc(64 * kStreamOps) : hipErrorUnknown;
        if (e == hipSuccess) e = hipMemcpy(h->stream_ops.p, ops, sizeof(ops), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            h->emit_tab.release();
            h->stream_ops.release();
            (void)hipStreamDestroy(h->own_stream);
            delete h;
            return fail(CALIB_E_HIP, std::string("stream record table: ") + (built ? hipGetErrorString(e) : "inconsistent"));
        }
    }
    if (const char* e = std::getenv("CALIB_LM_MODE")) h->lm_mode = std::atoi(e) ? CALIB_LM_TWO_KERNEL : CALIB_LM_FUSED;
    if (const char* e = std::getenv("CALIB_HEAD_LOADS")) h->head_loads = std::strcmp(e, "narrow") == 0 ? 1 : (std::strcmp(e, "wide") == 0 ? 2 : 0);
    if (const char* e = std::getenv("CALIB_ITEMS_PER_WAVE")) h->items_per_wave = std::max(0, std::min(16, std::atoi(e)));
    h->upd_lane_views = kUpdLaneViews;
    if (const char* e = std::getenv("CALIB_UPD_LANE_VIEWS")) h->upd_lane_views = std::max(1, std::atoi(e));
    h->upd_small_views = kUpdSmallViews;
    if (const char* e = std::getenv("CALIB_UPD_SMALL_VIEWS")) h->upd_small_views = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("CALIB_GRAM_FORM")) h->gram_form = std::strcmp(e, "tile") == 0 ? 1 : (std::strcmp(e, "block") == 0 ? 2 : 0);
    if (const char* e = std::getenv("CALIB_FUSED_STREAM")) h->stream_mode = std::atoi(e) > 0 ? 1 : (std::atoi(e) == 0 ? 0 : -1);
    if (const char* e = std::getenv("CALIB_STREAM_WAVES")) h->stream_waves_env = std::max(0, std::atoi(e));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            h->num_cus = cus;
    }
    *out_handle = h;
    return CALIB_OK;
}

namespace {
// Slot memory exported by handles of THIS process (one process driving several handles / GPUs): HIP IPC
// cannot open a handle in the process that made it, so calib_peer_connect looks here first.
struct LocalSlots { hipIpcMemHandle_t ipc; void* mem; int device; };
std::vector<LocalSlots> g_local_slots;
std::mutex g_local_slots_mutex;

void peer_release(calib_handle_s* h) {
    for (void* m : h->peer_open)
        if (m) (void)hipIpcCloseMemHandle(m);
    h->peer_open.clear();
    if (h->peer_mem) {
        std::lock_guard<std::mutex> lock(g_local_slots_mutex);
        for (size_t i = 0; i < g_local_slots.size(); ++i)
            if (g_local_slots[i].mem == h->peer_mem) { g_local_slots.erase(g_local_slots.begin() + (long)i); break; }
    }
    if (h->peer_mem) (void)hipFree(h->peer_mem);
    h->peer_mem = nullptr;
    h->peer_slots.release();
    h->peer_slot_host.clear();
    h->peer_flags.release();
    h->peer_connected = false;
    h->peer_world = 0;
}
}  // namespace

int calib_destroy(calib_handle_t h) {
    if (!h) return CALIB_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.commDestroy) { (void)g_rccl.commDestroy(h->comm); h->comm = nullptr; }
    peer_release(h);
    for (auto& e : h->ev) (void)hipEventDestroy(e);
    if (h->stage_ready)
        for (int t = 0; t < 4; ++t) {
            for (int b = 0; b < 2; ++b) {
                if (h->stage_ev[t][b]) (void)hipEventDestroy(h->stage_ev[t][b]);
            }
        }
    if (h->stage_pinned) (void)hipHostFree(h->stage_pinned);
    if (h->host_done) (void)hipHostFree(h->host_done);
    h->uv.release(); h->XY.release(); h->Z.release(); h->VC.release(); h->J.release();
    h->r.release(); h->y.release(); h->pt_view.release(); h->view_ext.release();
    h->item_n.release(); h->view_item0.release(); h->item_view.release(); h->item_pt0.release(); h->sse_part.release();
    h->emit_tab.release(); h->stream_ops.release();
    h->G[0].release(); h->G[1].release(); h->bpart.release(); h->part.release(); h->red_own.release();
    h->P[0].release(); h->P[1].release(); h->Peval.release(); h->trace.release();
    h->st.release(); h->st_eval.release(); h->rccl_test.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return CALIB_OK;
}

int calib_set_stream(calib_handle_t h, void* hip_stream, int use_own) {
    CHECK_H(h);
    SYNC_H(h);
    h->stream = use_own ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return CALIB_OK;
}

int calib_synchronize(calib_handle_t h) {
    CHECK_H(h);
    SYNC_H(h);
    return CALIB_OK;
}

int calib_set_lm_mode(calib_handle_t h, int mode) {
    CHECK_H(h);
    if (mode != CALIB_LM_FUSED && mode != CALIB_LM_TWO_KERNEL) return fail(CALIB_E_INVALID, "unknown LM mode");
    if (h->lm_active) return fail(CALIB_E_STATE, "cannot change the LM mode inside a run");
    h->lm_mode = mode;
    return CALIB_OK;
}

int calib_fused_form(calib_handle_t h, int* out_share, int* out_waves) {
    if (!h || !out_share || !out_waves) return fail(CALIB_E_INVALID, "null argument");
    *out_share = stream_rounds(h) ? h->stream_share : 0;
    *out_waves = stream_rounds(h) ? h->stream_waves : 0;
    return CALIB_OK;
}

int calib_num_shared(calib_handle_t h, int* out_L) {
    if (!h || !out_L) return fail(CALIB_E_INVALID, "null argument");
    *out_L = h->L;
    return CALIB_OK;
}

int calib_num_params(calib_handle_t h, int64_t* out_K) {
    if (!h || !out_K) return fail(CALIB_E_INVALID, "null argument");
    *out_K = numParams(h);
    return CALIB_OK;
}

}  // extern "C"
namespace {
int set_problem_impl(calib_handle_t h, int64_t num_views, const int64_t* view_offsets, const HostRows& sensor_uv,
                     const HostRows& model_xyz);
}
extern "C" {

int calib_set_problem(calib_handle_t h, int64_t num_views, const int64_t* view_offsets,
                      const double* sensor_uv, const double* model_xyz) {
    HostRows s, m;
    s.flat = sensor_uv; s.width = 2;
    m.flat = model_xyz; m.width = 3;
    return set_problem_impl(h, num_views, view_offsets, s, m);
}

int calib_set_problem_views(calib_handle_t h, int64_t num_views, const int64_t* view_offsets,
                            const double* const* sensor_uv_views, const double* const* model_xyz_views) {
    if (num_views < 0 || !view_offsets) return fail(CALIB_E_INVALID, "bad view_offsets");
    if (num_views > 0 && !model_xyz_views) return fail(CALIB_E_INVALID, "model_xyz_views is null");
    for (int64_t i = 0; i < num_views; ++i)
        if (view_offsets[i + 1] > view_offsets[i] && (!model_xyz_views[i] || (sensor_uv_views && !sensor_uv_views[i])))
            return fail(CALIB_E_INVALID, "a view with points has a null array");
    HostRows s, m;
    s.views = sensor_uv_views; s.offs = view_offsets; s.nviews = num_views; s.width = 2;
    m.views = model_xyz_views; m.offs = view_offsets; m.nviews = num_views; m.width = 3;
    return set_problem_impl(h, num_views, view_offsets, s, m);
}

}  // extern "C"
namespace {
int set_problem_impl(calib_handle_t h, int64_t num_views, const int64_t* view_offsets, const HostRows& sensor_uv,
                     const HostRows& model_xyz) {
    CHECK_H(h);
    if (num_views < 0 || !view_offsets) return fail(CALIB_E_INVALID, "bad view_offsets");
    if (view_offsets[0] != 0) return fail(CALIB_E_INVALID, "view_offsets[0] must be 0");
    for (int64_t i = 0; i < num_views; ++i)
        if (view_offsets[i + 1] < view_offsets[i])
            return fail(CALIB_E_INVALID, "view_offsets must be non-decreasing");
    const int64_t MN = view_offsets[num_views];
    if (MN > 0 && !model_xyz.present()) return fail(CALIB_E_INVALID, "model_xyz is null");
    if (num_views > 0x7fffffffLL / 8 || MN > (int64_t)1 << 40)
        return fail(CALIB_E_INVALID, "problem too large for one shard");
    SYNC_H(h);
    h->has_problem = false;
    h->lm_active = false;
    h->M = num_views;
    h->MN = MN;

    const bool timing = std::getenv("CALIB_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tmark = tnow();
    auto lap = [&](const char* what) { if (timing) { const double t = tnow(); std::fprintf(stderr, "  set_problem %-28s %.3f ms\n", what, t - tmark); tmark = t; } };
    // host side, O(views): compact (non-empty) view list, their point offsets, gram / fused work items.
    // Everything O(points) -- the AoS -> SoA split, the storage-type conversion, the point -> view index --
    // happens on the device from the caller's arrays uploaded as they are (pack_points_kernel).
    std::vector<int> view_ext, item_n, view_item0, item_view;
    std::vector<int64_t> item_pt0, voffs;
    view_item0.push_back(0);
    for (int64_t i = 0; i < num_views; ++i) {
        const int64_t a = view_offsets[i], b = view_offsets[i + 1];
        if (b == a) continue;
        const int cv = (int)view_ext.size();
        view_ext.push_back((int)i);
        voffs.push_back(a);
        for (int64_t p = a; p < b; p += kGramChunk) {
            item_pt0.push_back(p);
            item_n.push_back((int)std::min<int64_t>(kGramChunk, b - p));
            item_view.push_back(cv);
        }
        view_item0.push_back((int)item_pt0.size());
    }
    voffs.push_back(MN);
    h->nv = (int)view_ext.size();
    h->n_items = (int)item_pt0.size();
    // compact view of point p (points of the non-empty views are contiguous)
    auto viewOf = [&](int64_t p) { return (int)(std::upper_bound(voffs.begin(), voffs.end(), p) - voffs.begin()) - 1; };
    {   // waves per gram item from the mean points per item: a wave wants >= 2 trips of 16 points
        const double avg = h->n_items ? (double)MN / h->n_items : 0.0;
        h->gram_wpi = avg >= 128 ? 4 : (avg >= 64 ? 2 : 1);
        // fused kernel: one wave per item is fastest (measured c3: 84 us vs 102 us at 4) as long as
        // there are enough items to fill the chip; few big items are split over more waves
        h->fused_wpi = h->n_items >= 2048 ? 1 : h->gram_wpi;
        if (const char* e = std::getenv("CALIB_GRAM_WPI")) {      // tuning knob
            const int w = std::atoi(e);
            if (w == 1 || w == 2 || w == 4) h->gram_wpi = h->fused_wpi = w;
        }
    }
    h->n_tiles = (MN + kTile - 1) / kTile;
    int mv = 1;
    {   // views spanned by a 256-point tile (what the jacobian kernel stages in LDS): one sweep over the offsets
        int va = 0, vb = 0;
        for (int64_t t = 0; t < h->n_tiles; ++t) {
            const int64_t a = t * kTile, b = std::min<int64_t>(MN, a + kTile) - 1;
            while (voffs[(size_t)va + 1] <= a) ++va;
            if (vb < va) vb = va;
            while (voffs[(size_t)vb + 1] <= b) ++vb;
            mv = std::max(mv, vb - va + 1);
        }
    }
    {   // chunks of whole views, ~chunk_points each; tiles of a chunk start at the chunk's first point
        // Measured on MI355X (c3, 2 M points): chunks small enough for the 256 MiB Infinity Cache
        // do NOT make the J round trip cheaper (0.33 ms/iter at one chunk, 0.45 at 262 k points,
        // 1.1 at 65 k), so the chunk only bounds the J buffer: 64 M points = 17 GB at C = 16, fp64.
        int64_t target = (int64_t)1 << 26;
        if (const char* e = std::getenv("CALIB_CHUNK_POINTS")) target = std::max<int64_t>(1, std::atoll(e));
        h->chunks.clear();
        h->max_chunk_points = 0;
        int v = 0;
        while (v < h->nv) {
            calib_handle_s::Chunk c;
            c.p0 = item_pt0[(size_t)view_item0[(size_t)v]];
            c.item0 = view_item0[(size_t)v];
            int64_t p1 = c.p0;
            while (v < h->nv && (p1 - c.p0 < target)) {
                const int last = view_item0[(size_t)v + 1] - 1;
                p1 = item_pt0[(size_t)last] + item_n[(size_t)last];
                ++v;
            }
            c.p1 = p1;
            c.item1 = view_item0[(size_t)v];
            const bool whole = h->chunks.empty() && v >= h->nv;       // one chunk = the tiles counted above
            h->chunks.push_back(c);
            h->max_chunk_points = std::max(h->max_chunk_points, c.p1 - c.p0);
            if (!whole)
                for (int64_t a = c.p0; a < c.p1; a += kTile) {
                    const int64_t b = std::min<int64_t>(c.p1, a + kTile) - 1;
                    mv = std::max(mv, viewOf(b) - viewOf(a) + 1);
                }
        }
    }
    h->max_views_per_tile = mv;
    const int per = kSchurViewsPerBlock;
    h->schur_blocks = std::max(1, std::min(kMaxSchurBlocks, (h->nv + per - 1) / per));

    lap("host view/item lists");
    const size_t ts = tsize(h);
    HIP_TRY(h->uv.alloc((size_t)MN * 2 * ts));
    HIP_TRY(h->XY.alloc((size_t)MN * 2 * ts));
    HIP_TRY(h->Z.alloc((size_t)MN * ts));
    HIP_TRY(h->pt_view.alloc((size_t)MN));
    HIP_TRY(h->view_ext.alloc((size_t)h->nv));
    HIP_TRY(h->item_n.alloc((size_t)h->n_items));
    HIP_TRY(h->item_view.alloc((size_t)h->n_items));
    HIP_TRY(h->item_pt0.alloc((size_t)h->n_items));
    HIP_TRY(h->view_item0.alloc((size_t)h->nv + 1));
    HIP_TRY(h->VC.alloc((size_t)std::max(h->nv, 1) * kViewStride * ts));
    HIP_TRY(h->r.alloc((size_t)MN * 2 * ts));
    HIP_TRY(h->sse_part.alloc((size_t)std::max<int64_t>(h->n_tiles, 1)));
    HIP_TRY(h->st_eval.alloc(1));
    HIP_TRY(hipMemsetAsync(h->st_eval.p, 0, sizeof(LMState), h->stream));
    HIP_TRY(h->Peval.alloc((size_t)numParams(h)));

    lap("device allocations");
    auto upload = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    HIP_TRY(upload(h->view_ext.p, view_ext.data(), view_ext.size() * 4));
    {
        // uniform shards (every view fully detected: the usual case) need no item tables in the fused kernel
        int un = item_n.empty() ? 0 : item_n[0];
        for (size_t i = 0; i < item_n.size() && un > 0; ++i)
            if (item_n[i] != un || item_pt0[i] != (int64_t)i * un || item_view[i] != (int)i) un = 0;
        h->uniform_n = un;
    }
    {
        // Stream form of the fused kernel (kernels.hpp: fused_stream_kernel): uniform fp64 shards whose views are whole
        // 4-point groups and at least one batch long. One wave per wave slot of the chip (4 per SIMD), every wave the
        // same share of groups; a share is at least two views, so a view is cut by at most one wave start. By default
        // only where a wave gets two views or more anyway (below that, a view per wave fills the chip better).
        h->stream_share = h->stream_waves = 0;
        const int un = h->uniform_n;
        const bool can = h->dtype == CALIB_DTYPE_F64 && un >= 64 && (un & 3) == 0 && MN < ((int64_t)1 << 31) && h->nv >= 1;
        if (can && h->stream_mode != 0) {
            const int slots = h->stream_waves_env > 0 ? h->stream_waves_env : 4 * CALIB_STREAM_MIN_BLOCKS * h->num_cus;
            const int waves = std::max(1, std::min(slots, h->nv));
            if (h->stream_mode == 1 || h->nv >= 2 * slots) {
                const int64_t groups = (int64_t)h->nv * (un / 4);
                h->stream_share = (int)((groups + waves - 1) / waves);
                h->stream_waves = (int)((groups + h->stream_share - 1) / h->stream_share);
            }
        }
    }
    HIP_TRY(upload(h->item_n.p, item_n.data(), item_n.size() * 4));
    HIP_TRY(upload(h->item_view.p, item_view.data(), item_view.size() * 4));
    HIP_TRY(upload(h->item_pt0.p, item_pt0.data(), item_pt0.size() * 8));
    HIP_TRY(upload(h->view_item0.p, view_item0.data(), view_item0.size() * 4));
    if (MN > 0) {
        DevBuf<double> xyz_stage, uv_stage;
        DevBuf<int64_t> dvoffs;
        HIP_TRY(xyz_stage.alloc((size_t)MN * 3));
        HIP_TRY(dvoffs.alloc(voffs.size()));
        HIP_TRY(upload(dvoffs.p, voffs.data(), voffs.size() * 8));
        lap("small uploads + stage alloc");
        int rc = upload_staged(h, xyz_stage.p, model_xyz, (size_t)MN * 24);
        if (rc) return rc;
        lap("staged upload xyz");
        int uv_mode = 2;
        const double* uv_in = nullptr;
        if (sensor_uv.present() && h->dtype == CALIB_DTYPE_F64) {   // already in the device layout
            rc = upload_staged(h, h->uv.p, sensor_uv, (size_t)MN * 16);
            if (rc) return rc;
            uv_mode = 0;
        } else if (sensor_uv.present()) {
            HIP_TRY(uv_stage.alloc((size_t)MN * 2));
            rc = upload_staged(h, uv_stage.p, sensor_uv, (size_t)MN * 16);
            if (rc) return rc;
            uv_in = uv_stage.p;
            uv_mode = 1;
        }
        lap("staged upload uv");
        const unsigned blocks = (unsigned)((MN + 255) / 256);
        if (h->dtype == CALIB_DTYPE_F64)
            hipLaunchKernelGGL((pack_points_kernel<double>), dim3(blocks), dim3(256), 0, h->stream, xyz_stage.p, uv_in,
                               uv_mode, dvoffs.p, h->nv, MN, reinterpret_cast<double2*>(h->XY.p),
                               reinterpret_cast<double*>(h->Z.p), reinterpret_cast<double2*>(h->uv.p), h->pt_view.p);
        else
            hipLaunchKernelGGL((pack_points_kernel<float>), dim3(blocks), dim3(256), 0, h->stream, xyz_stage.p, uv_in,
                               uv_mode, dvoffs.p, h->nv, MN, reinterpret_cast<float2*>(h->XY.p),
                               reinterpret_cast<float*>(h->Z.p), reinterpret_cast<float2*>(h->uv.p), h->pt_view.p);
        LAUNCHED(h, "pack_points_kernel");
        SYNC_H(h);                   // the staging buffers go out of scope
        lap("pack kernel");
    }
    h->has_problem = true;
    return CALIB_OK;
}
}  // namespace
extern "C" {

int calib_eval(calib_handle_t h, const double* P, double* out_y, double* out_r, double* out_Jc,
               double* out_sse) {
    CHECK_H(h);
    int rc = need_problem(h);
    if (rc) return rc;
    if (!P) return fail(CALIB_E_INVALID, "P is null");
    const size_t ts = tsize(h);
    const int64_t MN = h->MN;
    const size_t MN4 = (size_t)((MN + 3) / 4 * 4);     // J is stored in groups of 4 points
    if (out_Jc) HIP_TRY(h->J.alloc(MN4 * h->C * 2 * ts));
    if (out_y) HIP_TRY(h->y.alloc((size_t)MN * 2 * ts));
    HIP_TRY(h->red_own.alloc((size_t)reduceSize(h->L)));
    HIP_TRY(hipMemcpyAsync(h->Peval.p, P, (size_t)numParams(h) * 8, hipMemcpyHostToDevice, h->stream));
    rc = launch_view_setup_any(h, h->Peval.p, nullptr, h->st_eval.p, 0);
    if (rc) return rc;
    rc = launch_jacobian(h, h->Peval.p, nullptr, h->st_eval.p, 0, out_Jc != nullptr, out_r != nullptr,
                         out_y != nullptr, true, 0, MN);
    if (rc) return rc;
    hipLaunchKernelGGL(sse_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->sse_part.p, h->n_tiles,
                       h->red_own.p);
    LAUNCHED(h, "sse_reduce_kernel");
    SYNC_H(h);
    if (out_sse) HIP_TRY(hipMemcpy(out_sse, h->red_own.p, 8, hipMemcpyDeviceToHost));

    auto fetch2 = [&](const void* dev, double* out) -> int {   // (MN,2) of T -> double
        if (h->dtype == CALIB_DTYPE_F64) {
            HIP_TRY(hipMemcpy(out, dev, (size_t)MN * 16, hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp((size_t)MN * 2);
            HIP_TRY(hipMemcpy(tmp.data(), dev, tmp.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < tmp.size(); ++i) out[i] = tmp[i];
        }
        return CALIB_OK;
    };
    if (out_y && MN) { rc = fetch2(h->y.p, out_y); if (rc) return rc; }
    if (out_r && MN) { rc = fetch2(h->r.p, out_r); if (rc) return rc; }
    if (out_Jc && MN) {
        // device layout jIndex(p, c)(du,dv) -> caller layout (MN,2,C)
        const int C = h->C;
        const size_t cnt = MN4 * C * 2;
        std::vector<double> tmp(cnt);
        if (h->dtype == CALIB_DTYPE_F64) {
            HIP_TRY(hipMemcpy(tmp.data(), h->J.p, cnt * 8, hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tf(cnt);
            HIP_TRY(hipMemcpy(tf.data(), h->J.p, cnt * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < cnt; ++i) tmp[i] = tf[i];
        }
        for (int64_t p = 0; p < MN; ++p)
            for (int c = 0; c < C; ++c) {
                out_Jc[(p * 2 + 0) * C + c] = tmp[(size_t)jIndex(p, c, C) * 2 + 0];
                out_Jc[(p * 2 + 1) * C + c] = tmp[(size_t)jIndex(p, c, C) * 2 + 1];
            }
    }
    return CALIB_OK;
}

int calib_lm_reduce_size(calib_handle_t h, int64_t* out_num_doubles) {
    if (!h || !out_num_doubles) return fail(CALIB_E_INVALID, "null argument");
    *out_num_doubles = reduceSize(h->L);
    return CALIB_OK;
}

int calib_lm_bind_reduce_buffer(calib_handle_t h, void* reduce_dev) {
    CHECK_H(h);
    SYNC_H(h);
    HIP_TRY(h->red_own.alloc((size_t)reduceSize(h->L)));
    h->red = reduce_dev ? reinterpret_cast<double*>(reduce_dev) : h->red_own.p;
    return CALIB_OK;
}

int calib_lm_begin(calib_handle_t h, const double* P0, int max_iters, double lam_init, double lam_min,
                   double lam_max, double err_min) {
    CHECK_H(h);
    int rc = need_problem(h);
    if (rc) return rc;
    if (!P0) return fail(CALIB_E_INVALID, "P0 is null");
    if (max_iters <= 0)
        return fail(CALIB_E_INVALID, "max_iters must be >= 1 (the reference raises UnboundLocalError "
                                     "for maxIters=0, src/calibrate.py:171)");
    // A shard WITHOUT views is legal (ranks of a sharded run may own none): it contributes zeros to the
    // reduce buffer and still takes every decision. Alone, it ends as the reference does on an empty system:
    // CALIB_E_SINGULAR from the L x L solve.
    if (h->nv != h->M)
        return fail(CALIB_E_SINGULAR, "a view without points makes J^T J + lambda diag(J^T J) singular");
    const size_t ts = tsize(h);
    const int64_t K = numParams(h);
    if (h->lm_mode == CALIB_LM_TWO_KERNEL)
        HIP_TRY(h->J.alloc((size_t)((h->max_chunk_points + 3) / 4 * 4) * h->C * 2 * ts));
    for (int b = 0; b < 2; ++b) {
        // fused_stream_kernel writes only the record entries the per-view kernels read: start from zeros
        const bool fresh = !h->G[b].p || h->G[b].n < (size_t)num_records(h) * kGStride;
        HIP_TRY(h->G[b].alloc((size_t)num_records(h) * kGStride));
        if (fresh) HIP_TRY(hipMemsetAsync(h->G[b].p, 0, h->G[b].n * 8, h->stream));
    }
    HIP_TRY(h->bpart.alloc(((size_t)std::max(h->n_items, 1) + h->chunks.size()) * kPartStride));   // <= 1 per item (+1 per chunk)
    HIP_TRY(h->part.alloc((size_t)2 * h->schur_blocks * variantSize(h->L)));
    HIP_TRY(h->red_own.alloc((size_t)reduceSize(h->L)));
    if (!h->red) h->red = h->red_own.p;
    HIP_TRY(h->P[0].alloc((size_t)K));
    HIP_TRY(h->P[1].alloc((size_t)K));
    HIP_TRY(h->st.alloc(2));
    HIP_TRY(h->trace.alloc((size_t)max_iters * (CALIB_TRACE_HEADER + h->L)));
    HIP_TRY(hipMemsetAsync(h->trace.p, 0, (size_t)max_iters * (CALIB_TRACE_HEADER + h->L) * 8, h->stream));
    HIP_TRY(hipMemcpyAsync(h->P[0].p, P0, (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->P[1].p, P0, (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    LMState s;
    std::memset(&s, 0, sizeof(s));
    s.lam = lam_init; s.lam_min = lam_min; s.lam_max = lam_max; s.err_min = err_min;
    if (!h->host_done) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(int), hipHostMallocMapped) == hipSuccess) h->host_done = static_cast<int*>(p);
        else (void)hipGetLastError();        // without the word the host falls back to synchronising checks
        void* dp = nullptr;
        if (h->host_done && hipHostGetDevicePointer(&dp, h->host_done, 0) == hipSuccess) h->host_done_dev = static_cast<int*>(dp);
        else (void)hipGetLastError();
    }
    if (h->host_done) {
        // a run that was begun and never ended (an exception between lmBegin and lmEnd) may still have update kernels
        // in flight that would set the word AFTER this reset: drain them first
        if (h->lm_active) SYNC_H(h);
        *h->host_done = 0;
        s.notify = h->host_done_dev;
    }
    s.cur = 1;            // round 0 evaluates the "candidate" buffer 0 == P0
    s.max_iters = max_iters;
    HIP_TRY(hipMemsetAsync(h->st.p, 0, 2 * sizeof(LMState), h->stream));
    HIP_TRY(hipMemcpyAsync(h->st.p, &s, sizeof(s), hipMemcpyHostToDevice, h->stream));
    SYNC_H(h);     // s and P0 are host stack / caller memory
    h->lm_active = true;
    h->lm_max_iters = max_iters;
    h->rounds_enqueued = 0;
    return CALIB_OK;
}

int calib_lm_local(calib_handle_t h) {
    CHECK_H(h);
    if (!h->lm_active) return fail(CALIB_E_STATE, "calib_lm_begin has not been called");
    LMState* st = st_cur(h);
    int rc = CALIB_OK;
    if (h->rounds_enqueued == 0) {      // later rounds: the update kernel already wrote the candidate's constants
        rc = launch_view_setup_any(h, h->P[0].p, h->P[1].p, st, 1);
        if (rc) return rc;
    }
    if (h->lm_mode == CALIB_LM_FUSED) {
        rc = launch_fused(h, st, 1);
        if (rc) return rc;
        return launch_schur_reduce(h, st, h->red);
    }
    h->n_bpart = 0;
    for (const auto& c : h->chunks) {
        rc = launch_jacobian(h, h->P[0].p, h->P[1].p, st, 1, true, true, false, false, c.p0, c.p1);
        if (rc) return rc;
        rc = launch_gram(h, st, 1, c.item0, c.item1, c.p0);
        if (rc) return rc;
    }
    return launch_schur_reduce(h, st, h->red);
}

int calib_lm_update(calib_handle_t h) {
    CHECK_H(h);
    if (!h->lm_active) return fail(CALIB_E_STATE, "calib_lm_begin has not been called");
    const int rc = launch_update_backsub(h);
    h->rounds_enqueued += 1;
    return rc;
}

namespace {
// after a synchronisation: did a peer exchange of this handle give up waiting for a rank?
int peer_fault_check(calib_handle_s* h) {
    if (!h->peer_connected) return CALIB_OK;
    int fault = 0;
    HIP_TRY(hipMemcpy(&fault, h->peer_flags.p, sizeof(int), hipMemcpyDeviceToHost));
    if (fault)
        return fail(CALIB_E_HIP, "peer exchange: a rank's contribution did not arrive before the deadline "
                                 "(the ranks no longer run in lockstep, or a peer died)");
    return CALIB_OK;
}
}  // namespace

int calib_lm_done(calib_handle_t h, int* out_done) {
    CHECK_H(h);
    if (!h->lm_active || !out_done) return fail(CALIB_E_STATE, "no LM run active");
    LMState s;
    HIP_TRY(hipMemcpyAsync(&s, st_cur(h), sizeof(s), hipMemcpyDeviceToHost, h->stream));
    SYNC_H(h);
    *out_done = s.done;
    return peer_fault_check(h);
}

int calib_lm_peek_trace(calib_handle_t h, int iter, double* out_row, int* out_iters) {
    CHECK_H(h);
    if (!h->lm_active || !out_row || !out_iters) return fail(CALIB_E_STATE, "no LM run active");
    if (iter < 0 || iter >= h->lm_max_iters) return fail(CALIB_E_INVALID, "trace row out of range");
    LMState s;
    const size_t w = (size_t)(CALIB_TRACE_HEADER + h->L);
    HIP_TRY(hipMemcpyAsync(&s, st_cur(h), sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(out_row, h->trace.p + (size_t)iter * w, w * 8, hipMemcpyDeviceToHost, h->stream));
    SYNC_H(h);
    *out_iters = s.iters;
    return CALIB_OK;
}

namespace {
int lm_run(calib_handle_t h, int rounds, int check_every, bool sharded);
}

int calib_lm_run(calib_handle_t h, int rounds, int check_every) { return lm_run(h, rounds, check_every, false); }

int calib_lm_run_sharded(calib_handle_t h, int rounds, int check_every) {
    if (h && !h->comm && !h->peer_connected)
        return fail(CALIB_E_STATE, "neither calib_peer_connect nor calib_rccl_init has been called");
    return lm_run(h, rounds, check_every, true);
}

namespace {
int lm_run(calib_handle_t h, int rounds, int check_every, bool sharded) {
    CHECK_H(h);
    if (!h->lm_active) return fail(CALIB_E_STATE, "calib_lm_begin has not been called");
    const bool peers = sharded && h->peer_connected;      // the reduce kernel sums over the ranks itself
    for (int i = 0; i < rounds; ++i) {
        h->exchange_round = peers;
        int rc = calib_lm_local(h);
        h->exchange_round = false;
        if (rc) return rc;
        if (sharded && !peers) {
            rc = calib_lm_allreduce(h);
            if (rc) return rc;
        }
        rc = calib_lm_update(h);
        if (rc) return rc;
        if (peers && h->host_done && *static_cast<volatile int*>(h->host_done) == 2)
            return fail(CALIB_E_HIP, "peer exchange: a rank's contribution did not arrive before the deadline "
                                     "(the ranks no longer run in lockstep, or a peer died); no further rounds are enqueued");
        if (check_every > 0 && !sharded && h->host_done) {
            // single shard: the device says so in host-visible memory when the loop is over -- no synchronisation, the
            // host simply stops enqueueing (rounds already in the queue exit at once)
            if (*static_cast<volatile int*>(h->host_done)) break;
        } else if (check_every > 0 && (i + 1) % check_every == 0 && i + 1 < rounds) {
            // sharded: every rank must enqueue the same rounds, so all of them look at the (replicated) flag at the
            // same round numbers
            int done = 0;
            rc = calib_lm_done(h, &done);
            if (rc) return rc;
            if (done) break;
        }
    }
    return CALIB_OK;
}
}  // namespace

int calib_rccl_load(const char* librccl_path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return CALIB_OK;
    if (!librccl_path) return fail(CALIB_E_INVALID, "librccl path is null");
    void* lib = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(CALIB_E_HIP, std::string("dlopen librccl: ") + dlerror());
    RcclApi a;
    a.lib = lib;
    a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(dlsym(lib, "ncclCommInitRank"));
    a.allReduce = reinterpret_cast<decltype(a.allReduce)>(dlsym(lib, "ncclAllReduce"));
    a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(lib, "ncclCommDestroy"));
    a.commAbort = reinterpret_cast<decltype(a.commAbort)>(dlsym(lib, "ncclCommAbort"));
    a.getErrorString = reinterpret_cast<decltype(a.getErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!a.getUniqueId || !a.commInitRank || !a.allReduce || !a.commDestroy || !a.commAbort) {
        (void)dlclose(lib);
        return fail(CALIB_E_HIP, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclCommAbort");
    }
    g_rccl = a;
    return CALIB_OK;
}

int calib_rccl_unique_id(void* out_id128) {
    if (!g_rccl.lib) return fail(CALIB_E_STATE, "calib_rccl_load has not been called");
    if (!out_id128) return fail(CALIB_E_INVALID, "null argument");
    RcclId id;
    const int rc = g_rccl.getUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(out_id128, id.internal, sizeof(id.internal));
    return CALIB_OK;
}

int calib_rccl_init(calib_handle_t h, int nranks, int rank, const void* id128) {
    return calib_rccl_init_deadline(h, nranks, rank, id128, 120.0);
}

int calib_rccl_init_deadline(calib_handle_t h, int nranks, int rank, const void* id128, double timeout_s) {
    CHECK_H(h);
    if (!g_rccl.lib) return fail(CALIB_E_STATE, "calib_rccl_load has not been called");
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(CALIB_E_INVALID, "bad communicator arguments");
    if (h->comm) return fail(CALIB_E_STATE, "this handle already has a communicator");
    // ncclCommInitRank blocks until every rank has joined; a peer that never calls it would hang this thread
    // for good. It runs on a helper thread; past the deadline the caller gets an error and the helper, should
    // it ever return, destroys the communicator it made.
    struct Job { RcclId id; int nranks, rank, device; void* comm = nullptr; int rc = 0; std::atomic<bool> abandoned{false}; };
    auto job = std::make_shared<Job>();
    std::memcpy(job->id.internal, id128, sizeof(job->id.internal));
    job->nranks = nranks; job->rank = rank; job->device = h->device;
    std::promise<void> donePromise;
    std::future<void> done = donePromise.get_future();
    std::thread([job, p = std::move(donePromise)]() mutable {
        (void)hipSetDevice(job->device);
        job->rc = g_rccl.commInitRank(&job->comm, job->nranks, job->id, job->rank);
        if (job->abandoned.load() && job->rc == 0 && job->comm) (void)g_rccl.commAbort(job->comm);
        p.set_value();
    }).detach();
    if (done.wait_for(std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 120.0)) != std::future_status::ready) {
        job->abandoned.store(true);
        return fail(CALIB_E_HIP, "ncclCommInitRank did not return before the deadline (a peer rank is missing)");
    }
    if (job->rc) return rccl_fail("ncclCommInitRank", job->rc);
    h->comm = job->comm;
    h->comm_ranks = nranks;
    h->comm_rank = rank;
    return CALIB_OK;
}

int calib_rccl_shutdown(calib_handle_t h) {
    CHECK_H(h);
    if (!h->comm) return CALIB_OK;
    (void)hipStreamSynchronize(h->stream);
    const int rc = g_rccl.commDestroy(h->comm);
    h->comm = nullptr;
    h->comm_ranks = 0;
    if (rc) return rccl_fail("ncclCommDestroy", rc);
    return CALIB_OK;
}

int calib_rccl_selftest(calib_handle_t h, double timeout_s) {
    CHECK_H(h);
    if (!h->comm) return fail(CALIB_E_STATE, "calib_rccl_init has not been called");
    // rank r contributes (r + 1, 1, 2^r, 4): the sums must come back as (n (n + 1) / 2, n, 2^n - 1, 4 n).
    // The operand lives in the handle: a collective that timed out may still refer to it.
    const int n = h->comm_ranks, r = h->comm_rank;
    HIP_TRY(h->rccl_test.alloc(4));
    const double mine[4] = {r + 1.0, 1.0, std::ldexp(1.0, r), 4.0};
    const double want[4] = {0.5 * n * (n + 1.0), (double)n, std::ldexp(1.0, n) - 1.0, 4.0 * n};
    HIP_TRY(hipMemcpy(h->rccl_test.p, mine, sizeof(mine), hipMemcpyHostToDevice));
    auto giveUp = [&]() {                   // the communicator cannot be trusted: abort it, never destroy it
        (void)g_rccl.commAbort(h->comm);
        h->comm = nullptr;
        h->comm_ranks = 0;
    };
    int rc = g_rccl.allReduce(h->rccl_test.p, h->rccl_test.p, 4, kNcclDouble, kNcclSum, h->comm, h->stream);
    if (rc) { giveUp(); return rccl_fail("ncclAllReduce (self-test)", rc); }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s > 0 ? timeout_s : 30.0);
    for (;;) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) { giveUp(); return fail(CALIB_E_HIP, hipGetErrorString(e)); }
        if (std::chrono::steady_clock::now() > deadline) {
            giveUp();
            return fail(CALIB_E_HIP, "in-library all-reduce self-test timed out");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    double got[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(got, h->rccl_test.p, sizeof(got), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i)
        if (got[i] != want[i]) {
            giveUp();
            return fail(CALIB_E_HIP, "in-library all-reduce self-test returned wrong sums");
        }
    return CALIB_OK;
}

int calib_lm_allreduce(calib_handle_t h) {
    CHECK_H(h);
    if (!h->lm_active) return fail(CALIB_E_STATE, "calib_lm_begin has not been called");
    if (!h->comm) return fail(CALIB_E_STATE, "calib_rccl_init has not been called");
    const int rc = g_rccl.allReduce(h->red, h->red, (size_t)reduceSize(h->L), kNcclDouble, kNcclSum, h->comm, h->stream);
    if (rc) return rccl_fail("ncclAllReduce", rc);
    return CALIB_OK;
}

// ---- peer exchange over xGMI ----------------------------------------------------------------
int calib_peer_prepare(calib_handle_t h, int nranks, int rank, void* out_handle64) {
    CHECK_H(h);
    static_assert(sizeof(hipIpcMemHandle_t) == CALIB_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    if (!out_handle64 || nranks < 1 || nranks > kPeerMaxRanks || rank < 0 || rank >= nranks)
        return fail(CALIB_E_INVALID, "bad peer exchange arguments (1 <= nranks <= 64, 0 <= rank < nranks)");
    if (h->peer_mem) return fail(CALIB_E_STATE, "this handle already takes part in a peer exchange");
    if (h->lm_active) return fail(CALIB_E_STATE, "cannot set up a peer exchange inside an LM run");
    // Cells are written by other GPUs while this one polls them: the memory must not be held in this
    // device's L2 (uncached; fine-grained where the runtime has no uncached pool).
    const size_t bytes = (size_t)nranks * 2 * kPeerStride * 2 * sizeof(unsigned long long);
    void* mem = nullptr;
    hipError_t e = hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) return fail(CALIB_E_HIP, std::string("peer slot memory: ") + hipGetErrorString(e));
    h->peer_mem = mem;
    h->peer_world = nranks;
    h->peer_rank = rank;
    auto undo = [&](int code, const std::string& msg) { peer_release(h); return fail(code, msg); };
    e = hipMemset(mem, 0, bytes);          // epoch 0 is never sent
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return undo(CALIB_E_HIP, std::string("peer slot memory: ") + hipGetErrorString(e));
    hipIpcMemHandle_t ipc;
    e = hipIpcGetMemHandle(&ipc, mem);
    if (e != hipSuccess) return undo(CALIB_E_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
    std::memcpy(out_handle64, &ipc, sizeof(ipc));
    {
        std::lock_guard<std::mutex> lock(g_local_slots_mutex);
        g_local_slots.push_back(LocalSlots{ipc, mem, h->device});
    }
    return CALIB_OK;
}

int calib_peer_connect(calib_handle_t h, const void* handles, double timeout_s) {
    CHECK_H(h);
    if (!handles) return fail(CALIB_E_INVALID, "null argument");
    if (!h->peer_mem) return fail(CALIB_E_STATE, "calib_peer_prepare has not been called");
    if (h->peer_connected) return fail(CALIB_E_STATE, "peers are already connected");
    const int n = h->peer_world;
    std::vector<unsigned long long*> slots((size_t)n, nullptr);
    h->peer_open.assign((size_t)n, nullptr);
    for (int r = 0; r < n; ++r) {
        if (r == h->peer_rank) { slots[(size_t)r] = static_cast<unsigned long long*>(h->peer_mem); continue; }
        hipIpcMemHandle_t ipc;
        std::memcpy(&ipc, static_cast<const char*>(handles) + (size_t)r * sizeof(ipc), sizeof(ipc));
        void* m = nullptr;
        hipError_t pe = hipSuccess;
        {
            std::lock_guard<std::mutex> lock(g_local_slots_mutex);
            for (const LocalSlots& ls : g_local_slots)
                if (std::memcmp(&ls.ipc, &ipc, sizeof(ipc)) == 0) {
                    if (ls.device != h->device) {
                        pe = hipDeviceEnablePeerAccess(ls.device, 0);
                        if (pe == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); pe = hipSuccess; }
                    }
                    m = ls.mem;
                }
        }
        if (pe != hipSuccess) {
            (void)hipGetLastError();
            for (void*& o : h->peer_open) { if (o) (void)hipIpcCloseMemHandle(o); o = nullptr; }
            return fail(CALIB_E_HIP, "hipDeviceEnablePeerAccess (device of rank " + std::to_string(r) + "): " + hipGetErrorString(pe));
        }
        if (m) { slots[(size_t)r] = static_cast<unsigned long long*>(m); continue; }
        const hipError_t e = hipIpcOpenMemHandle(&m, ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (void*& o : h->peer_open) { if (o) (void)hipIpcCloseMemHandle(o); o = nullptr; }
            return fail(CALIB_E_HIP, "hipIpcOpenMemHandle (slot memory of rank " + std::to_string(r) + "): " + hipGetErrorString(e));
        }
        h->peer_open[(size_t)r] = m;
        slots[(size_t)r] = static_cast<unsigned long long*>(m);
    }
    h->peer_slot_host = slots;
    HIP_TRY(h->peer_slots.alloc((size_t)n));
    HIP_TRY(h->peer_flags.alloc(2));
    HIP_TRY(hipMemcpy(h->peer_slots.p, slots.data(), (size_t)n * sizeof(slots[0]), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->peer_flags.p, 0, 2 * sizeof(int)));
    h->peer_timeout_s = timeout_s > 0 ? timeout_s : 60.0;
    h->peer_epoch = 0;
    h->peer_connected = true;
    return CALIB_OK;
}

int calib_peer_selftest(calib_handle_t h, int rounds, double timeout_s) {
    CHECK_H(h);
    if (!h->peer_connected) return fail(CALIB_E_STATE, "calib_peer_connect has not been called");
    if (rounds < 1) rounds = 1;
    const double keep = h->peer_timeout_s;
    h->peer_timeout_s = timeout_s > 0 ? timeout_s : 10.0;       // each spin of the test gives up after this long
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(peer_selftest_kernel, dim3(kPeerStride), dim3(64), 0, h->stream, next_exchange(h), r,
                           h->peer_flags.p + 1);
        LAUNCHED(h, "peer_selftest_kernel");
    }
    h->peer_timeout_s = keep;
    SYNC_H(h);
    int flags[2] = {0, 0};
    HIP_TRY(hipMemcpy(flags, h->peer_flags.p, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags[0] || flags[1]) {
        HIP_TRY(hipMemset(h->peer_flags.p, 0, sizeof(flags)));
        return fail(CALIB_E_HIP, flags[0] ? "peer exchange self-test: a rank's contribution did not arrive in time"
                                          : "peer exchange self-test: wrong sums (" + std::to_string(flags[1]) + " elements)");
    }
    return CALIB_OK;
}

int calib_peer_shutdown(calib_handle_t h) {
    CHECK_H(h);
    if (h->lm_active) return fail(CALIB_E_STATE, "cannot drop the peer exchange inside an LM run");
    (void)hipStreamSynchronize(h->stream);
    peer_release(h);
    return CALIB_OK;
}

int calib_lm_end(calib_handle_t h, double* P_out, double* out_sse, int* out_iters, double* out_trace) {
    CHECK_H(h);
    if (!h->lm_active) return fail(CALIB_E_STATE, "no LM run active");
    SYNC_H(h);
    LMState s;
    HIP_TRY(hipMemcpy(&s, st_cur(h), sizeof(s), hipMemcpyDeviceToHost));
    h->lm_active = false;
    const int prc = peer_fault_check(h);
    if (prc) return prc;
    if (s.error == CALIB_E_SINGULAR)
        return fail(CALIB_E_SINGULAR, "Singular matrix: damped normal equations are not invertible");
    // after the bootstrap round cur points at the current parameters
    const int cur = s.round == 0 ? 0 : s.cur;
    if (P_out) HIP_TRY(hipMemcpy(P_out, h->P[cur].p, (size_t)numParams(h) * 8, hipMemcpyDeviceToHost));
    if (out_sse) *out_sse = s.last_err;
    if (out_iters) *out_iters = s.iters;
    if (out_trace && s.iters > 0)
        HIP_TRY(hipMemcpy(out_trace, h->trace.p, (size_t)s.iters * (CALIB_TRACE_HEADER + h->L) * 8,
                          hipMemcpyDeviceToHost));
    return CALIB_OK;
}

int calib_refine(calib_handle_t h, double* P_inout, int max_iters, double lam_init, double lam_min,
                 double lam_max, double err_min, double* out_sse, int* out_iters, double* out_trace) {
    int rc = calib_lm_begin(h, P_inout, max_iters, lam_init, lam_min, lam_max, err_min);
    if (rc) return rc;
    rc = calib_lm_run(h, max_iters + 1, 8);
    if (rc) return rc;
    return calib_lm_end(h, P_inout, out_sse, out_iters, out_trace);
}

int calib_lm_step_delta(calib_handle_t h, const double* P, double lambda, double* out_delta) {
    if (!out_delta) return fail(CALIB_E_INVALID, "out_delta is null");
    int rc = calib_lm_begin(h, P, 1, lambda, 0.0, INFINITY, -INFINITY);
    if (rc) return rc;
    rc = calib_lm_run(h, 1, 0);     // bootstrap round: evaluates P, solves, writes P + delta
    if (rc) return rc;
    SYNC_H(h);
    LMState s;
    HIP_TRY(hipMemcpy(&s, st_cur(h), sizeof(s), hipMemcpyDeviceToHost));
    h->lm_active = false;
    const int prc = peer_fault_check(h);
    if (prc) return prc;
    if (s.error == CALIB_E_SINGULAR)
        return fail(CALIB_E_SINGULAR, "Singular matrix: damped normal equations are not invertible");
    const int64_t K = numParams(h);
    std::vector<double> cand((size_t)K);
    HIP_TRY(hipMemcpy(cand.data(), h->P[s.cur ^ 1].p, (size_t)K * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < K; ++i) out_delta[i] = cand[(size_t)i] - P[i];
    return CALIB_OK;
}

int calib_normal_eq(calib_handle_t h, const double* P, double* out_B, double* out_E, double* out_V,
                    double* out_g) {
    // bootstrap round with lambda = 0: variant A of the reduce buffer carries sum B and g_c,
    // the per-view Gram blocks carry E_i, V_i, g_i.
    int rc = calib_lm_begin(h, P, 1, 0.0, 0.0, INFINITY, -INFINITY);
    if (rc) return rc;
    rc = calib_lm_local(h);
    if (rc) return rc;
    SYNC_H(h);
    h->lm_active = false;
    const int L = h->L;
    std::vector<double> red((size_t)reduceSize(L));
    HIP_TRY(hipMemcpy(red.data(), h->red, red.size() * 8, hipMemcpyDeviceToHost));
    if (out_B) std::memcpy(out_B, red.data(), (size_t)L * L * 8);
    if (out_g) std::memcpy(out_g, red.data() + 2 * L * L, (size_t)L * 8);
    if (out_E || out_V || out_g) {
        std::vector<double> G((size_t)num_records(h) * kGStride);
        std::vector<int> vi0((size_t)h->nv + 1);
        const StreamMap sm = stream_map(h);
        HIP_TRY(hipMemcpy(G.data(), h->G[0].p, G.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(vi0.data(), h->view_item0.p, vi0.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < h->nv; ++v) {       // nv == M here (calib_lm_begin checked)
            double blk[kGStride];
            std::fill(blk, blk + kGStride, 0.0);
            for (int it = vi0[v]; it < vi0[v + 1]; ++it)
                for (int i = 0; i < kGStride; ++i) blk[i] += G[(size_t)it * kGStride + i];
            if (sm.share > 0) {                 // stream form: the part of the view summed by a second wave
                const int64_t first = (int64_t)v * sm.n4, w = (first + sm.n4 - 1) / sm.share;
                if (w * sm.share > first)
                    for (int i = 0; i < kGStride; ++i) blk[i] += G[(size_t)(sm.nv + w) * kGStride + i];
            }
            for (int a = 0; a < 6; ++a) {
                if (out_g) out_g[L + 6 * (int64_t)v + a] = blk[gvSlot(L, a)];
                for (int b = 0; b < 6; ++b)     // V is symmetric; the record holds its lower triangle (g_v rides in the upper)
                    if (out_V) out_V[((int64_t)v * 6 + a) * 6 + b] = blk[kGRows + (a >= b ? a * 16 + L + b : b * 16 + L + a)];
                for (int c = 0; c < L; ++c)
                    if (out_E) out_E[((int64_t)v * L + c) * 6 + a] = blk[kGRows + a * 16 + c];
            }
        }
    }
    return CALIB_OK;
}

#ifdef CALIB_STREAM_STAMPS
// diagnostic build only (tools/diag/): the stamped waves' rows of g_sstamps (kSStampSlots values each), then clears
int calib_debug_stream_stamps(unsigned long long* out, int max_waves) {
    std::vector<unsigned long long> hst((size_t)calib::kSStampWaves * calib::kSStampSlots);
    (void)hipMemcpyFromSymbol(hst.data(), HIP_SYMBOL(calib::g_sstamps), hst.size() * 8);
    const int nw = std::min(max_waves, calib::kSStampWaves);
    if (out) std::memcpy(out, hst.data(), (size_t)nw * calib::kSStampSlots * 8);
    std::fill(hst.begin(), hst.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(calib::g_sstamps), hst.data(), hst.size() * 8);
    return 0;
}
#endif

int calib_distort_points(int model, int64_t n, const double* x_norm, const double* k, double* out_xd) {
    if (n < 0 || (n > 0 && (!x_norm || !out_xd)) || !k) return fail(CALIB_E_INVALID, "null argument");
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE)
        return fail(CALIB_E_INVALID, "unknown distortion model");
    if (n == 0) return CALIB_OK;
    const int nk = model == CALIB_MODEL_RADTAN ? 5 : 4;
    DevBuf<double> dx, dk, dout;
    HIP_TRY(dx.alloc((size_t)n * 2)); HIP_TRY(dk.alloc(nk)); HIP_TRY(dout.alloc((size_t)n * 2));
    HIP_TRY(hipMemcpy(dx.p, x_norm, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dk.p, k, (size_t)nk * 8, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (model == CALIB_MODEL_RADTAN)
        hipLaunchKernelGGL((distort_points_kernel<kRadtan>), dim3(blocks), dim3(256), 0, 0, dx.p, dk.p, n, dout.p);
    else
        hipLaunchKernelGGL((distort_points_kernel<kFisheye>), dim3(blocks), dim3(256), 0, 0, dx.p, dk.p, n, dout.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_xd, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost);
    dx.release(); dk.release(); dout.release();
    if (e != hipSuccess) return fail(CALIB_E_HIP, hipGetErrorString(e));
    return CALIB_OK;
}

int calib_project_with_distortion(int model, int64_t n, const double* A, const double* cam_xyz,
                                  const double* k, double* out_uv) {
    if (n < 0 || (n > 0 && (!cam_xyz || !out_uv)) || !k || !A) return fail(CALIB_E_INVALID, "null argument");
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE)
        return fail(CALIB_E_INVALID, "unknown distortion model");
    if (n == 0) return CALIB_OK;
    const int nk = model == CALIB_MODEL_RADTAN ? 5 : 4;
    DevBuf<double> dA, dc, dk, dout;
    HIP_TRY(dA.alloc(9)); HIP_TRY(dc.alloc((size_t)n * 3)); HIP_TRY(dk.alloc(nk)); HIP_TRY(dout.alloc((size_t)n * 2));
    HIP_TRY(hipMemcpy(dA.p, A, 72, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.p, cam_xyz, (size_t)n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dk.p, k, (size_t)nk * 8, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (model == CALIB_MODEL_RADTAN)
        hipLaunchKernelGGL((project_cam_kernel<kRadtan>), dim3(blocks), dim3(256), 0, 0, dA.p, dc.p, dk.p, n, dout.p);
    else
        hipLaunchKernelGGL((project_cam_kernel<kFisheye>), dim3(blocks), dim3(256), 0, 0, dA.p, dc.p, dk.p, n, dout.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_uv, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost);
    dA.release(); dc.release(); dk.release(); dout.release();
    if (e != hipSuccess) return fail(CALIB_E_HIP, hipGetErrorString(e));
    return CALIB_OK;
}

namespace {

int check_views(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv, const double* model_xyz) {
    if (num_views < 0 || !view_offsets) return fail(CALIB_E_INVALID, "null argument");
    if (num_views == 0) return CALIB_OK;
    const int64_t MN = view_offsets[num_views];
    if (view_offsets[0] != 0 || MN < 0 || (MN > 0 && (!sensor_uv || !model_xyz)))
        return fail(CALIB_E_INVALID, "bad view_offsets / point arrays");
    for (int64_t i = 0; i < num_views; ++i)
        if (view_offsets[i + 1] < view_offsets[i]) return fail(CALIB_E_INVALID, "view_offsets must be non-decreasing");
    return CALIB_OK;
}

int use_device(int device_id) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(CALIB_E_HIP, "no such HIP device (no CPU fallback)");
    HIP_TRY(hipSetDevice(device_id));
    return CALIB_OK;
}

// DLT and / or LM polish of every view's homography
int homography_pipeline(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv,
                        const double* model_xyz, double* H, bool dlt, int refine_iters, int device_id) {
    if (!H) return fail(CALIB_E_INVALID, "null argument");
    int rc = check_views(num_views, view_offsets, sensor_uv, model_xyz);
    if (rc || num_views == 0) return rc;
    rc = use_device(device_id);
    if (rc) return rc;
    const int64_t MN = view_offsets[num_views];
    std::vector<double> xy((size_t)MN * 2);
    for (int64_t p = 0; p < MN; ++p) { xy[2 * p] = model_xyz[3 * p]; xy[2 * p + 1] = model_xyz[3 * p + 1]; }
    DevBuf<int64_t> doffs;
    DevBuf<double> duv, dxy, dH;
    hipError_t e = doffs.alloc((size_t)num_views + 1);
    if (e == hipSuccess) e = duv.alloc((size_t)std::max<int64_t>(MN, 1) * 2);
    if (e == hipSuccess) e = dxy.alloc((size_t)std::max<int64_t>(MN, 1) * 2);
    if (e == hipSuccess) e = dH.alloc((size_t)num_views * 9);
    if (e == hipSuccess) e = hipMemcpy(doffs.p, view_offsets, ((size_t)num_views + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && MN) e = hipMemcpy(duv.p, sensor_uv, (size_t)MN * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && MN) e = hipMemcpy(dxy.p, xy.data(), (size_t)MN * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && !dlt) e = hipMemcpy(dH.p, H, (size_t)num_views * 72, hipMemcpyHostToDevice);
    const unsigned blocks = (unsigned)((num_views + 15) / 16);
    if (e == hipSuccess && dlt) {
        hipLaunchKernelGGL(dlt_kernel, dim3(blocks), dim3(256), 0, 0, doffs.p, reinterpret_cast<const double2*>(duv.p),
                           reinterpret_cast<const double2*>(dxy.p), num_views, dH.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess && refine_iters > 0) {
        hipLaunchKernelGGL(homography_lm_kernel, dim3(blocks), dim3(256), 0, 0, doffs.p,
                           reinterpret_cast<const double2*>(duv.p), reinterpret_cast<const double2*>(dxy.p),
                           num_views, refine_iters, dH.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(H, dH.p, (size_t)num_views * 72, hipMemcpyDeviceToHost);
    doffs.release(); duv.release(); dxy.release(); dH.release();
    if (e != hipSuccess) return fail(CALIB_E_HIP, hipGetErrorString(e));
    return CALIB_OK;
}

}  // namespace

int calib_refine_homographies(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv,
                              const double* model_xyz, double* H_inout, int max_iters, int device_id) {
    return homography_pipeline(num_views, view_offsets, sensor_uv, model_xyz, H_inout, false, max_iters, device_id);
}

int calib_estimate_homographies(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv,
                                const double* model_xyz, double* H_out, int refine_iters, int device_id) {
    if (num_views > 0 && view_offsets)
        for (int64_t i = 0; i < num_views; ++i)
            if (view_offsets[i + 1] - view_offsets[i] < 4)
                return fail(CALIB_E_INVALID, "a homography needs at least 4 point correspondences per view");
    return homography_pipeline(num_views, view_offsets, sensor_uv, model_xyz, H_out, true, refine_iters, device_id);
}

int calib_homography_jacobian(int64_t n, const double* h9, const double* model_xyz, double* out_J, int device_id) {
    if (n < 0 || !h9 || (n > 0 && (!model_xyz || !out_J))) return fail(CALIB_E_INVALID, "null argument");
    if (n == 0) return CALIB_OK;
    int rc = use_device(device_id);
    if (rc) return rc;
    DevBuf<double> dh, dx, dJ;
    HIP_TRY(dh.alloc(9));
    HIP_TRY(dx.alloc((size_t)n * 3));
    HIP_TRY(dJ.alloc((size_t)n * 18));
    HIP_TRY(hipMemcpy(dh.p, h9, 72, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dx.p, model_xyz, (size_t)n * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(homography_jacobian_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dh.p, dx.p, n, dJ.p);
    LAUNCHED(static_cast<calib_handle_s*>(nullptr), "homography_jacobian_kernel");
    HIP_TRY(hipMemcpy(out_J, dJ.p, (size_t)n * 18 * 8, hipMemcpyDeviceToHost));
    return CALIB_OK;
}

int calib_compute_extrinsics(int64_t num_views, const double* A, const double* H, double* W_out, int device_id) {
    if (num_views < 0 || !A || (num_views > 0 && (!H || !W_out))) return fail(CALIB_E_INVALID, "null argument");
    if (num_views == 0) return CALIB_OK;
    int rc = use_device(device_id);
    if (rc) return rc;
    // A = [[a, g, uc], [0, b, vc], [0, 0, 1]] (src/calibrate.py:252-256): closed-form inverse
    const double a = A[0], g = A[1], uc = A[2], b = A[4], vc = A[5];
    if (!(a != 0.0) || !(b != 0.0)) return fail(CALIB_E_SINGULAR, "Singular matrix: intrinsic matrix is not invertible");
    const double Ainv[9] = {1.0 / a, -g / (a * b), (g * vc - uc * b) / (a * b), 0.0, 1.0 / b, -vc / b, 0.0, 0.0, 1.0};
    DevBuf<double> dA, dH, dW;
    hipError_t e = dA.alloc(9);
    if (e == hipSuccess) e = dH.alloc((size_t)num_views * 9);
    if (e == hipSuccess) e = dW.alloc((size_t)num_views * 16);
    if (e == hipSuccess) e = hipMemcpy(dA.p, Ainv, 72, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dH.p, H, (size_t)num_views * 72, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(extrinsics_kernel, dim3((unsigned)((num_views + 255) / 256)), dim3(256), 0, 0, dA.p, dH.p,
                           num_views, dW.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(W_out, dW.p, (size_t)num_views * 128, hipMemcpyDeviceToHost);
    dA.release(); dH.release(); dW.release();
    if (e != hipSuccess) return fail(CALIB_E_HIP, hipGetErrorString(e));
    return CALIB_OK;
}

int calib_distortion_normal_equations(int model, int64_t num_views, const int64_t* view_offsets,
                                      const double* sensor_uv, const double* model_xyz, const double* A,
                                      const double* W, double* out_DtD, double* out_Dtd, int device_id) {
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE) return fail(CALIB_E_INVALID, "unknown distortion model");
    if (!A || !out_DtD || !out_Dtd || (num_views > 0 && !W)) return fail(CALIB_E_INVALID, "null argument");
    int rc = check_views(num_views, view_offsets, sensor_uv, model_xyz);
    if (rc) return rc;
    const int nk = model == CALIB_MODEL_RADTAN ? 5 : 4;
    const int ns = nk * (nk + 1) / 2 + nk;
    std::fill(out_DtD, out_DtD + nk * nk, 0.0);
    std::fill(out_Dtd, out_Dtd + nk, 0.0);
    const int64_t MN = num_views ? view_offsets[num_views] : 0;
    if (MN == 0) return CALIB_OK;
    rc = use_device(device_id);
    if (rc) return rc;
    std::vector<double> xy((size_t)MN * 2), z((size_t)MN);
    std::vector<int> pv((size_t)MN);
    for (int64_t v = 0; v < num_views; ++v)
        for (int64_t p = view_offsets[v]; p < view_offsets[v + 1]; ++p) pv[(size_t)p] = (int)v;
    for (int64_t p = 0; p < MN; ++p) { xy[2 * p] = model_xyz[3 * p]; xy[2 * p + 1] = model_xyz[3 * p + 1]; z[p] = model_xyz[3 * p + 2]; }
    const int blocks = (int)std::min<int64_t>(1024, (MN + 255) / 256);
    DevBuf<double> dA, dW, duv, dxy, dz, dpart;
    DevBuf<int> dpv;
    hipError_t e = dA.alloc(9);
    if (e == hipSuccess) e = dW.alloc((size_t)num_views * 16);
    if (e == hipSuccess) e = duv.alloc((size_t)MN * 2);
    if (e == hipSuccess) e = dxy.alloc((size_t)MN * 2);
    if (e == hipSuccess) e = dz.alloc((size_t)MN);
    if (e == hipSuccess) e = dpv.alloc((size_t)MN);
    if (e == hipSuccess) e = dpart.alloc((size_t)blocks * ns);
    if (e == hipSuccess) e = hipMemcpy(dA.p, A, 72, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dW.p, W, (size_t)num_views * 128, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(duv.p, sensor_uv, (size_t)MN * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dxy.p, xy.data(), (size_t)MN * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dz.p, z.data(), (size_t)MN * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dpv.p, pv.data(), (size_t)MN * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (model == CALIB_MODEL_RADTAN)
            hipLaunchKernelGGL((distortion_normal_kernel<kRadtan>), dim3(blocks), dim3(256), 0, 0, dA.p, dW.p, dpv.p,
                               reinterpret_cast<const double2*>(duv.p), reinterpret_cast<const double2*>(dxy.p), dz.p, MN, dpart.p);
        else
            hipLaunchKernelGGL((distortion_normal_kernel<kFisheye>), dim3(blocks), dim3(256), 0, 0, dA.p, dW.p, dpv.p,
                               reinterpret_cast<const double2*>(duv.p), reinterpret_cast<const double2*>(dxy.p), dz.p, MN, dpart.p);
        e = hipGetLastError();
    }
    std::vector<double> part((size_t)blocks * ns);
    if (e == hipSuccess) e = hipMemcpy(part.data(), dpart.p, part.size() * 8, hipMemcpyDeviceToHost);
    dA.release(); dW.release(); duv.release(); dxy.release(); dz.release(); dpv.release(); dpart.release();
    if (e != hipSuccess) return fail(CALIB_E_HIP, hipGetErrorString(e));
    std::vector<double> sum((size_t)ns, 0.0);
    for (int bidx = 0; bidx < blocks; ++bidx)
        for (int j = 0; j < ns; ++j) sum[(size_t)j] += part[(size_t)bidx * ns + j];
    int idx = 0;
    for (int a2 = 0; a2 < nk; ++a2)
        for (int b2 = a2; b2 < nk; ++b2) { out_DtD[a2 * nk + b2] = sum[(size_t)idx]; out_DtD[b2 * nk + a2] = sum[(size_t)idx]; ++idx; }
    for (int a2 = 0; a2 < nk; ++a2) out_Dtd[a2] = sum[(size_t)idx++];
    return CALIB_OK;
}

namespace {
int num_distortion(int model) { return model == CALIB_MODEL_RADTAN ? 5 : 4; }
}

int calib_compose_params(int model, int64_t num_views, const double* A, const double* W, const double* k,
                         double* P_out, int device_id) {
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE) return fail(CALIB_E_INVALID, "unknown distortion model");
    if (num_views < 0 || !A || !k || !P_out || (num_views > 0 && !W)) return fail(CALIB_E_INVALID, "null argument");
    const int nk = num_distortion(model), L = 5 + nk;
    // shared part: (alpha, beta, gamma, uc, vc, k...) from A = [[alpha, gamma, uc], [0, beta, vc], [0, 0, 1]]
    P_out[0] = A[0]; P_out[1] = A[4]; P_out[2] = A[1]; P_out[3] = A[2]; P_out[4] = A[5];
    for (int j = 0; j < nk; ++j) P_out[5 + j] = k[j];
    if (num_views == 0) return CALIB_OK;
    int rc = use_device(device_id);
    if (rc) return rc;
    DevBuf<double> dW, dP;
    HIP_TRY(dW.alloc((size_t)num_views * 16));
    HIP_TRY(dP.alloc((size_t)L + 6 * (size_t)num_views));
    HIP_TRY(hipMemcpy(dW.p, W, (size_t)num_views * 128, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(compose_views_kernel, dim3((unsigned)((num_views + 255) / 256)), dim3(256), 0, 0, dW.p, num_views, L, dP.p);
    LAUNCHED(static_cast<calib_handle_s*>(nullptr), "compose_views_kernel");
    HIP_TRY(hipMemcpy(P_out + L, dP.p + L, (size_t)num_views * 48, hipMemcpyDeviceToHost));
    return CALIB_OK;
}

int calib_decompose_params(int model, int64_t num_views, const double* P, double* A_out, double* W_out,
                           double* k_out, int device_id) {
    if (model != CALIB_MODEL_RADTAN && model != CALIB_MODEL_FISHEYE) return fail(CALIB_E_INVALID, "unknown distortion model");
    if (num_views < 0 || !P || (num_views > 0 && !W_out)) return fail(CALIB_E_INVALID, "null argument");
    const int nk = num_distortion(model), L = 5 + nk;
    if (A_out) {
        const double a[9] = {P[0], P[2], P[3], 0.0, P[1], P[4], 0.0, 0.0, 1.0};
        std::memcpy(A_out, a, sizeof(a));
    }
    if (k_out) for (int j = 0; j < nk; ++j) k_out[j] = P[5 + j];
    if (num_views == 0) return CALIB_OK;
    int rc = use_device(device_id);
    if (rc) return rc;
    DevBuf<double> dW, dP;
    HIP_TRY(dW.alloc((size_t)num_views * 16));
    HIP_TRY(dP.alloc((size_t)L + 6 * (size_t)num_views));
    HIP_TRY(hipMemcpy(dP.p, P, ((size_t)L + 6 * (size_t)num_views) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(decompose_views_kernel, dim3((unsigned)((num_views + 255) / 256)), dim3(256), 0, 0, dP.p, num_views, L, dW.p);
    LAUNCHED(static_cast<calib_handle_s*>(nullptr), "decompose_views_kernel");
    HIP_TRY(hipMemcpy(W_out, dW.p, (size_t)num_views * 128, hipMemcpyDeviceToHost));
    return CALIB_OK;
}

int calib_refine_awk(calib_handle_t h, double* A_inout, double* W_inout, double* k_inout, int max_iters,
                     double lam_init, double lam_min, double lam_max, double err_min, double* out_sse,
                     int* out_iters, double* out_trace) {
    CHECK_H(h);
    int rc = need_problem(h);
    if (rc) return rc;
    if (!A_inout || !k_inout || (h->M > 0 && !W_inout)) return fail(CALIB_E_INVALID, "null argument");
    std::vector<double> P((size_t)numParams(h));
    rc = calib_compose_params(h->model, h->M, A_inout, W_inout, k_inout, P.data(), h->device);
    if (rc) return rc;
    rc = calib_refine(h, P.data(), max_iters, lam_init, lam_min, lam_max, err_min, out_sse, out_iters, out_trace);
    if (rc) return rc;
    return calib_decompose_params(h->model, h->M, P.data(), A_inout, W_inout, k_inout, h->device);
}

int calib_profile_enable(calib_handle_t h, int on) {
    CHECK_H(h);
    SYNC_H(h);
    if (on && h->ev.empty()) {
        h->ev.resize(kEventPool);
        h->ev_kind.assign(kEventPool / 2, 0);
        for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
    }
    h->prof = on != 0;
    h->prof_stride = on > 1 ? on : 1;
    h->prof_seen[0] = h->prof_seen[1] = h->prof_seen[2] = 0;
    h->ev_used = 0;
    return CALIB_OK;
}

int calib_profile_read(calib_handle_t h, int which, double* out_total_ms, int64_t* out_launches) {
    CHECK_H(h);
    if (!out_total_ms || !out_launches) return fail(CALIB_E_INVALID, "null argument");
    SYNC_H(h);
    double total = 0.0;
    int64_t count = 0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        if (h->ev_kind[i / 2] != which) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        total += ms;
        count += 1;
    }
    *out_total_ms = total;
    *out_launches = count;
    return CALIB_OK;
}

}  // extern "C"
