/*
 * calib_lm.h -- C-ABI of the MI355X (gfx950) Levenberg-Marquardt refinement engine.
 *
 * Drop-in boundary for the nonlinear stage of pvphan/camera-calibration. The
 * reference has no FFI layer; its seam is the duck-typed Python surface of
 *   Calibrator.refineCalibrationParameters   src/calibrate.py:117-171
 *   ProjectionJacobian.compute               src/jacobian.py:48-85
 *   Calibrator.projectAllPoints              src/calibrate.py:190-197
 *   Calibrator._computeReprojectionError     src/calibrate.py:178-188
 *   DistortionModel.projectWithDistortion    src/distortion.py:42-59
 * The Python host in camera-calibration_amd/ binds these entry points with
 * ctypes (INTEGRATION.md shows the stub a maintainer adds to src/calibrate.py).
 *
 * Conventions
 *  - Every function returns 0 on success, <0 on error (CALIB_E_*);
 *    calib_last_error() returns a thread-local message. No C++ exception
 *    crosses the boundary.
 *  - All pointer arguments are caller-owned HOST memory unless the name ends
 *    in _dev. The library owns all device memory of a handle.
 *  - One handle = one GPU = one host thread (one process per GPU; shards of a
 *    multi-GPU job are separate handles in separate processes). Every entry point that takes a handle makes
 *    the handle's device the calling thread's current HIP device (hipSetDevice) and leaves it so.
 *  - Parameter vector P (fp64, length K = L + 6*M), exactly the reference's
 *    (src/calibrate.py:199-229):
 *      P = (alpha, beta, gamma, uc, vc, k[0..|k|),  then per view i:
 *           rho_x, rho_y, rho_z [DEGREES], t_x, t_y, t_z)
 *    L = 10 radial-tangential (k1,k2,p1,p2,k3), 9 fisheye (k1..k4).
 *  - Views are a CSR over points: view i owns points [view_offsets[i], view_offsets[i+1]).
 */
#ifndef CALIB_LM_H
#define CALIB_LM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct calib_handle_s* calib_handle_t;

enum { CALIB_MODEL_RADTAN = 0, CALIB_MODEL_FISHEYE = 1 };   /* src/main.py:28-33 */
enum { CALIB_DTYPE_F64 = 0, CALIB_DTYPE_F32 = 1 };          /* storage/eval type of points, J, r;
                                                               normal equations + solve are always fp64 */
enum {
    CALIB_OK = 0,
    CALIB_E_INVALID = -1,   /* bad argument / shape (reference: ValueError, src/mathutils.py:102-105) */
    CALIB_E_HIP = -2,       /* HIP runtime error, no device, kernel failure */
    CALIB_E_SINGULAR = -3,  /* damped normal equations singular (reference: numpy LinAlgError, src/calibrate.py:152) */
    CALIB_E_STATE = -4      /* call sequence error (e.g. lm_local before lm_begin) */
};

/* trace row written per LM iteration (backs shouldPrint, src/calibrate.py:158-159,269-274):
 *   [0] iter  [1] err(P) [2] err(P+delta) [3] lambda used  [4] accepted (0/1)
 *   [5 .. 5+L) the shared parameters of P before the update                       */
#define CALIB_TRACE_HEADER 5

int         calib_version(void);
const char* calib_last_error(void);
int         calib_device_count(int* out_count);

int calib_create(int model, int dtype, int device_id, calib_handle_t* out_handle);
int calib_destroy(calib_handle_t h);

/* Launch all work of this handle on an existing HIP stream (e.g. the stream a
 * torch.distributed all-reduce is ordered against). use_own != 0: back to the handle's own
 * (non-blocking) stream; otherwise hip_stream is used as given, NULL being the HIP default
 * stream (which is what torch.cuda.current_stream().cuda_stream is unless the caller changed it). */
int calib_set_stream(calib_handle_t h, void* hip_stream, int use_own);
/* Wait until everything enqueued on the handle's stream has run (hipStreamSynchronize); an asynchronous fault of an
 * earlier launch is reported here with the names of the kernels enqueued since the last successful wait. */
int calib_synchronize(calib_handle_t h);

/* Upload the correspondences once (replaces getSensorPoints' vstack, src/calibrate.py:277-282).
 * sensor_uv: (MN,2) row-major as numpy, may be NULL (projection-only use);
 * model_xyz: (MN,3) row-major. Packs to SoA on the device. */
int calib_set_problem(calib_handle_t h, int64_t num_views, const int64_t* view_offsets,
                      const double* sensor_uv, const double* model_xyz);
/* The same upload from PER-VIEW arrays, as the reference's callers hold them -- allDetections is a list of
 * (sensorPoints (N_i,2), modelPoints (N_i,3)) pairs (src/calibrate.py:117-118) that getSensorPoints stacks with np.vstack on
 * every call (src/calibrate.py:277-282). sensor_uv_views[i] / model_xyz_views[i] point at view i's C-contiguous rows
 * (view_offsets[i+1] - view_offsets[i] of them; NULL allowed for an empty view; sensor_uv_views itself may be NULL).
 * Nothing is stacked on the host: the staged upload gathers from the views straight into its pinned buffers. */
int calib_set_problem_views(calib_handle_t h, int64_t num_views, const int64_t* view_offsets,
                            const double* const* sensor_uv_views, const double* const* model_xyz_views);

/* How an LM round forms the per-view normal equations:
 *   CALIB_LM_FUSED       one kernel evaluates the 2 x C Jacobian blocks and contracts them with
 *                        MFMA without writing them to HBM (default; fastest);
 *   CALIB_LM_TWO_KERNEL  the jacobian kernel materialises the compact J in HBM (the layout
 *                        calib_eval returns) and the gram kernel reads it back.
 * Both produce the same blocks up to summation order. Every sum is fixed-order, so a run is bitwise reproducible for a
 * given shard, mode and device. Bit-identity ACROSS modes, shard splits and devices holds for the one-view-per-wave
 * forms only: the stream form of the fused kernel (calib_fused_form) sums a view that a wave start cuts in two parts,
 * and where the cuts fall depends on the launch width (CU count, CALIB_STREAM_WAVES) -- "same sums, order fixed per
 * (shard shape, launch width)", 1e-13 relative between forms. CALIB_FUSED_STREAM=0 keeps the one-view-per-wave forms
 * for callers who need bit-reproducibility across devices. */
enum { CALIB_LM_FUSED = 0, CALIB_LM_TWO_KERNEL = 1 };
int calib_set_lm_mode(calib_handle_t h, int mode);
/* Which form of the fused kernel the loaded problem's rounds run in (decided by calib_set_problem from the shard's
 * shape; CALIB_FUSED_STREAM=0|1 and CALIB_STREAM_WAVES override it for tests and tuning).
 *   *out_share  > 0: stream form -- uniform fp64 shards (every view n points, n % 4 == 0, n >= 64, at least two views
 *               per wave slot of the chip): the shard's 4-point groups are dealt out to `*out_waves` waves in equal
 *               shares of `*out_share` groups, batches run across view boundaries, and a view cut by a wave start has
 *               a second (overflow) record; 0: one view item per wave (or several whole items per wave). */
int calib_fused_form(calib_handle_t h, int* out_share, int* out_waves);

int calib_num_shared(calib_handle_t h, int* out_L);          /* L                          */
int calib_num_params(calib_handle_t h, int64_t* out_K);       /* K = L + 6*num_views        */

/* One evaluation at P. Any output may be NULL.
 *   out_y   (MN,2)    projection            -> Calibrator.projectAllPoints
 *   out_r   (MN,2)    sensor - projection   -> residual of src/calibrate.py:151
 *   out_Jc  (MN,2,C)  compact Jacobian, C = L+6: per point rows (du, dv), columns
 *                     [shared L | the point's own view's 6]  -> ProjectionJacobian.compute
 *   out_sse           sum of squared residual norms -> _computeReprojectionError           */
int calib_eval(calib_handle_t h, const double* P, double* out_y, double* out_r,
               double* out_Jc, double* out_sse);

/* Block-arrow normal equations at P (J^T J and J^T r of src/calibrate.py:146,152):
 *   out_B (L,L), out_E (M,L,6), out_V (M,6,6), out_g (K).  Any may be NULL. */
int calib_normal_eq(calib_handle_t h, const double* P, double* out_B, double* out_E,
                    double* out_V, double* out_g);

/* delta = (J^T J + lambda diag(J^T J))^-1 J^T r at P (src/calibrate.py:152), solved on the
 * device through the Schur complement of the 6x6 view blocks. out_delta (K). */
int calib_lm_step_delta(calib_handle_t h, const double* P, double lambda, double* out_delta);

/* The whole refinement loop of src/calibrate.py:143-171 on the device.
 * P_inout (K): start point in, refined parameters out.
 * out_sse: the reference's return value = err(P) evaluated BEFORE the last update.
 * out_trace: max_iters * (CALIB_TRACE_HEADER + L) doubles or NULL.
 * max_iters <= 0 is CALIB_E_INVALID (the reference raises UnboundLocalError). */
int calib_refine(calib_handle_t h, double* P_inout, int max_iters,
                 double lam_init, double lam_min, double lam_max, double err_min,
                 double* out_sse, int* out_iters, double* out_trace);

/* Parameter vector (de)composition behind the boundary: Calibrator._composeParameterVector /
 * _decomposeParameterVector (src/calibrate.py:199-267). A (3,3) row-major, W (M,4,4) board poses in the
 * camera frame, k (5 radtan | 4 fisheye) <-> P (L + 6 M). Rotations become Euler angles in DEGREES with the
 * gimbal-lock branches of rotationMatrixToEuler (src/mathutils.py:13-33) and back through
 * eulerToRotationMatrix (src/mathutils.py:36-51); one device thread per view. Needs no handle.
 * calib_refine_awk = compose -> calib_refine -> decompose on the handle's problem: the signature of
 * Calibrator.refineCalibrationParameters (src/calibrate.py:117) for a C caller. */
int calib_compose_params(int model, int64_t num_views, const double* A, const double* W, const double* k,
                         double* P_out, int device_id);
int calib_decompose_params(int model, int64_t num_views, const double* P, double* A_out, double* W_out,
                           double* k_out, int device_id);
int calib_refine_awk(calib_handle_t h, double* A_inout, double* W_inout, double* k_inout, int max_iters,
                     double lam_init, double lam_min, double lam_max, double err_min,
                     double* out_sse, int* out_iters, double* out_trace);

/* ---- stepping form of the same loop (multi-GPU shards, benchmarks) ----------------
 * One LM round = calib_lm_local (per-shard kernels, fills the reduce buffer)
 *              -> [all-reduce(sum) of the reduce buffer across shards]
 *              -> calib_lm_update (accept/reject, lambda, solve, back-substitute).
 * Round 0 bootstraps (evaluates P0); rounds 1..max_iters are the LM iterations.
 * All calls only enqueue work on the handle's stream. */
int calib_lm_begin(calib_handle_t h, const double* P0, int max_iters,
                   double lam_init, double lam_min, double lam_max, double err_min);
int calib_lm_reduce_size(calib_handle_t h, int64_t* out_num_doubles);
/* Use a caller-owned DEVICE buffer of calib_lm_reduce_size doubles as the reduce buffer
 * (e.g. the storage of a torch tensor handed to torch.distributed.all_reduce). NULL restores
 * the library's own buffer. */
int calib_lm_bind_reduce_buffer(calib_handle_t h, void* reduce_dev);
int calib_lm_local(calib_handle_t h);
int calib_lm_update(calib_handle_t h);
/* rounds x (local, update) without a collective (single shard) -- also on a handle that owns a
 * communicator: calib_refine, calib_lm_step_delta and calib_normal_eq, which are built on it, never take part
 * in a collective. If check_every > 0 the host reads the device's done flag every check_every rounds and
 * stops early. */
int calib_lm_run(calib_handle_t h, int rounds, int check_every);
/* rounds x (local, exchange, update) with the exchange done by the library: summed inside the reduce kernel
 * over xGMI when peers are connected (calib_peer_connect), else an ncclAllReduce (calib_rccl_init). Every rank
 * must make the same call. CALIB_E_STATE when the handle has neither. */
int calib_lm_run_sharded(calib_handle_t h, int rounds, int check_every);

/* ---- peer exchange over xGMI (optional; the fastest form of the one exchange) -----------------------
 * The sum over the ranks happens INSIDE the kernel that finishes a shard's reduce buffer: each wave stores its
 * element point-to-point into every other rank's slot memory and polls its own for theirs (kernels.hpp,
 * "peer exchange"); no collective is launched, no stream is handed over, and every rank adds in rank order,
 * so the reduced system -- hence every accept/reject decision -- is bitwise identical everywhere. It replaces
 * the all-reduce torch.distributed / RCCL would launch per LM round. One process per GPU, ranks on one node:
 *   every rank: calib_peer_prepare(h, nranks, rank, handle64)       allocates its slot memory, exports it (HIP IPC)
 *   all-gather the CALIB_PEER_HANDLE_BYTES-byte handles by any means (rank order)
 *   every rank: calib_peer_connect(h, handles, timeout_s)            maps the peers' slot memory
 *   every rank: calib_peer_selftest(h, rounds, timeout_s)            exchanges known values, checks the sums
 * then calib_lm_run_sharded runs whole rounds. A rank that waits longer than timeout_s (<= 0: 60 s) for a
 * contribution stops waiting: the step is rejected, and calib_lm_done / calib_lm_end (and the self-test)
 * return CALIB_E_HIP -- the kernels never spin without bound. Handles of one exchange must call
 * calib_lm_run_sharded in lockstep (same number of rounds; they stop together because the done flag is derived
 * from the identical sums). calib_lm_run and everything built on it stay single-shard on a connected handle.
 * calib_peer_shutdown (or calib_destroy) unmaps and frees. nranks <= 64.
 * Exercised with one rank per process on ONE GPU (tests/test_gpu_multiproc.py); callers should keep the
 * self-test and fall back to calib_rccl_* / their own all-reduce when it fails on any rank. */
#define CALIB_PEER_HANDLE_BYTES 64
int calib_peer_prepare(calib_handle_t h, int nranks, int rank, void* out_handle64);
int calib_peer_connect(calib_handle_t h, const void* handles, double timeout_s);
int calib_peer_selftest(calib_handle_t h, int rounds, double timeout_s);
int calib_peer_shutdown(calib_handle_t h);

/* ---- in-library all-reduce (optional) -------------------------------------------------------------
 * The ONE exchange of the sharded loop can also be issued by the library itself, as ncclAllReduce on
 * the engine's own stream (no hand-off to another framework's stream, and calib_lm_run then drives
 * whole rounds -- local, all-reduce, update -- from C). RCCL is not linked: calib_rccl_load dlopens the
 * librccl.so the process already uses (path given by the caller, e.g. the one PyTorch ships) and
 * resolves ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclCommAbort.
 *   rank 0: calib_rccl_unique_id(id) -> broadcast the 128 bytes by any means -> every rank:
 *   calib_rccl_init(h, nranks, rank, id)   (collective)  -> calib_rccl_selftest(h, timeout_s)
 * calib_rccl_init blocks until every rank has joined, for at most 120 s (calib_rccl_init_deadline: a
 * caller-chosen limit); past the deadline it returns CALIB_E_HIP and the handle has no communicator.
 * The self-test all-reduces a rank-dependent vector and checks the sums, polling the stream against the
 * deadline; on a wrong answer or a timeout it ABORTS the communicator (ncclCommAbort) and returns
 * CALIB_E_HIP, and the caller falls back to its own all-reduce (calib_lm_bind_reduce_buffer). The collective is
 * never implicit: calib_lm_run_sharded runs whole rounds (local, all-reduce, update) and calib_lm_allreduce is
 * that step alone for callers that drive the three steps themselves; calib_lm_run and everything built on it
 * stay single-shard. Exercised with one rank per process on one GPU (tests/test_gpu_multiproc.py); more than
 * one GPU has not been available to this build. */
int calib_rccl_load(const char* librccl_path);
int calib_rccl_unique_id(void* out_id128);
int calib_rccl_init(calib_handle_t h, int nranks, int rank, const void* id128);
int calib_rccl_init_deadline(calib_handle_t h, int nranks, int rank, const void* id128, double timeout_s);
int calib_rccl_selftest(calib_handle_t h, double timeout_s);
int calib_rccl_shutdown(calib_handle_t h);
int calib_lm_allreduce(calib_handle_t h);
int calib_lm_done(calib_handle_t h, int* out_done);           /* synchronises */
/* Synchronise and copy trace row `iter` (CALIB_TRACE_HEADER + L doubles) of the running loop;
 * out_iters = LM iterations executed so far (rows >= that are not written yet). */
int calib_lm_peek_trace(calib_handle_t h, int iter, double* out_row, int* out_iters);
/* Synchronise, return refined P (K), reference-style sse, iterations executed, trace. */
int calib_lm_end(calib_handle_t h, double* P_out, double* out_sse, int* out_iters,
                 double* out_trace);

/* Numeric forward model on caller points (no problem needed):
 *   calib_distort_points            DistortionModel.distortPoints          src/distortion.py:78-108,198-220
 *   calib_project_with_distortion   DistortionModel.projectWithDistortion  src/distortion.py:42-59
 * x_norm (N,2) normalised points; cam_xyz (N,3) camera-frame points; A (3,3) row-major. */
int calib_distort_points(int model, int64_t n, const double* x_norm, const double* k,
                         double* out_xd);
int calib_project_with_distortion(int model, int64_t n, const double* A, const double* cam_xyz,
                                  const double* k, double* out_uv);

/* Per-view homography polish of the initialisation stage: Calibrator._refineHomographies
 * (src/calibrate.py:60-111, HomographyJacobian src/jacobian.py:88-121), all views in one launch.
 * H_inout (M,3,3) row-major: DLT homographies in, LM-refined (H[2,2] = 1) out. model_xyz (MN,3):
 * only X, Y are used. max_iters = 20 in the reference. */
int calib_refine_homographies(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv,
                              const double* model_xyz, double* H_inout, int max_iters, int device_id);

/* HomographyJacobian.compute (src/jacobian.py:88-121): the (2N, 9) Jacobian of the projection of the
 * model points (X, Y, 1) through H = h.reshape(3,3) with respect to h, rows (u_j, v_j) interleaved.
 * h9 (9), model_xyz (N,3): only X, Y are used; out_J (2N,9) row-major. */
int calib_homography_jacobian(int64_t n, const double* h9, const double* model_xyz, double* out_J,
                              int device_id);

/* ---- closed-form initialisation stage on the device (Calibrator.estimateCalibrationParameters,
 * src/calibrate.py:41-58). The 6-unknown intrinsics fit and the <= 5-unknown distortion solve stay on
 * the host; everything that is per view / per point runs here. ------------------------------------
 * calib_estimate_homographies: normalised DLT (src/linearcalibrate.py:7-90) for every view, followed by
 *   refine_iters LM iterations (0 = DLT only). H_out (M,3,3), H[2,2] = 1.
 * calib_compute_extrinsics: world-to-camera poses from A (3,3) and the homographies
 *   (src/linearcalibrate.py:306-371). W_out (M,4,4).
 * calib_distortion_normal_equations: D^T D (n,n) and D^T Ddot (n) of the linear distortion estimate
 *   D k = Ddot (src/distortion.py:110-191, 222-271), n = 5 radtan / 4 fisheye; W (M,4,4). */
int calib_estimate_homographies(int64_t num_views, const int64_t* view_offsets, const double* sensor_uv,
                                const double* model_xyz, double* H_out, int refine_iters, int device_id);
int calib_compute_extrinsics(int64_t num_views, const double* A, const double* H, double* W_out, int device_id);
int calib_distortion_normal_equations(int model, int64_t num_views, const int64_t* view_offsets,
                                      const double* sensor_uv, const double* model_xyz, const double* A,
                                      const double* W, double* out_DtD, double* out_Dtd, int device_id);

/* HIP-event timing of the dominant kernels over the rounds enqueued since the last
 * calib_profile_enable(h, on). on = 0: off; on = 1: every launch is bracketed by an event pair;
 * on = N > 1: every N-th launch of each kernel (an event pair keeps a launch from being dispatched
 * back to back with its neighbours, so timing every launch slows the loop it measures).
 * which: 0 = jacobian kernel, 1 = J^T J (MFMA) kernel, 2 = fused jacobian + J^T J kernel;
 * out_launches = launches timed, out_total_ms = their summed duration. */
int calib_profile_enable(calib_handle_t h, int on);
int calib_profile_read(calib_handle_t h, int which, double* out_total_ms, int64_t* out_launches);

#ifdef __cplusplus
}
#endif
#endif /* CALIB_LM_H */
