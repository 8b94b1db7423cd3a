"""RefineEngine: one GPU shard of the LM refinement engine (thin wrapper over the C-ABI).

Holds the correspondences of a set of views resident in HBM and evaluates
projection / residual / Jacobian / normal equations / the whole LM loop there.
"""
import ctypes

import numpy as np

from . import _native as nat

LAMBDA_INITIAL = 1e-3      # Calibrator._λinitial      src/calibrate.py:13
LAMBDA_MIN = 1e-10         # Calibrator._λmin          src/calibrate.py:14
LAMBDA_MAX = 1e+10         # Calibrator._λmax          src/calibrate.py:15
PT_ERROR_MIN = 1e-12       # Calibrator._Pt_error_min  src/calibrate.py:16

MODEL_IDS = {"radtan": nat.MODEL_RADTAN, "fisheye": nat.MODEL_FISHEYE}
NUM_SHARED = {nat.MODEL_RADTAN: 10, nat.MODEL_FISHEYE: 9}
DTYPE_IDS = {"f64": nat.DTYPE_F64, "fp64": nat.DTYPE_F64, "float64": nat.DTYPE_F64,
             "f32": nat.DTYPE_F32, "fp32": nat.DTYPE_F32, "float32": nat.DTYPE_F32}


def _stack(arrays, width):
    """list of (N_i, width) arrays -> (counts (M,) int64, stacked (sum N_i, width) float64, C-contiguous).
    One pass over the list when every element already is a float64 (N, width) ndarray -- what the reference's own
    callers hand over -- and ONE np.concatenate (a single allocation, a memcpy per view); anything else (lists,
    other dtypes) is converted view by view first. Round 3 walked the list three times with asarray / reshape per
    view and a vstack of temporaries: 10-12 s for 100 000 views, against ~0.3 s for the loop below."""
    M = len(arrays)
    if M == 0:
        return np.zeros(0, dtype=np.int64), np.empty((0, width))
    fast = True
    for a in arrays:
        if type(a) is not np.ndarray or a.dtype != np.float64 or a.ndim != 2 or a.shape[1] != width:
            fast = False
            break
    if not fast:
        conv = []
        for i, a in enumerate(arrays):
            a = np.asarray(a)
            if a.ndim != 2 or a.shape[1] != width:
                raise ValueError(f"view {i}: expected shape (N, {width}), got {a.shape}")
            conv.append(np.asarray(a, dtype=np.float64))
        arrays = conv
    counts = np.fromiter((a.shape[0] for a in arrays), dtype=np.int64, count=M)
    return counts, np.concatenate(arrays, axis=0)


_fastpack = False        # the optional CPython helper lib/_fastpack*.so (csrc/fastpack.c): False = not looked for yet


def _fastpackModule():
    global _fastpack
    if _fastpack is False:
        import glob
        import importlib.machinery
        import importlib.util
        import os
        _fastpack = None
        for path in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_fastpack*.so")):
            try:
                loader = importlib.machinery.ExtensionFileLoader("_fastpack", path)
                spec = importlib.util.spec_from_loader("_fastpack", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _fastpack = mod
                break
            except Exception:       # noqa: BLE001 -- built for another interpreter: numpy stacks instead
                _fastpack = None
    return _fastpack


def viewPointers(allDetections):
    """list of (sensorPoints (N_i,2), modelPoints (N_i,3)) -> (CSR offsets int64[M+1], sensor addresses uint64[M],
    model addresses uint64[M]) WITHOUT stacking anything: what calib_set_problem_views gathers from. None when the
    helper module is not built or a view is not a C-contiguous float64 array of the right width (the caller then
    stacks with packDetections). The addresses are valid while the caller keeps allDetections alive."""
    fp = _fastpackModule()
    if fp is None or len(allDetections) == 0:
        return None
    a = fp.view_pointers(allDetections, 0, 2)
    b = fp.view_pointers(allDetections, 1, 3) if a is not None else None
    if a is None or b is None:
        return None
    if not np.array_equal(a[0], b[0]):
        i = int(np.flatnonzero(a[0] != b[0])[0])
        raise ValueError(f"view {i}: expected sensor (N,2) and model (N,3), got ({a[0][i]}, 2) and ({b[0][i]}, 3)")
    offs = np.zeros(len(allDetections) + 1, dtype=np.int64)
    np.cumsum(a[0], out=offs[1:])
    return offs, a[1], b[1]


def packDetections(allDetections):
    """list of (sensorPoints (N_i,2), modelPoints (N_i,3)) -> CSR offsets + stacked arrays.
    The stacking is getSensorPoints' vstack (src/calibrate.py:277-282) done once."""
    M = len(allDetections)
    for i, d in enumerate(allDetections):
        if len(d) != 2:
            raise ValueError(f"view {i}: expected a (sensorPoints, modelPoints) pair")
    ns, sensor = _stack([d[0] for d in allDetections], 2)
    nm, model = _stack([d[1] for d in allDetections], 3)
    if M and not np.array_equal(ns, nm):
        i = int(np.flatnonzero(ns != nm)[0])
        raise ValueError(f"view {i}: expected sensor (N,2) and model (N,3), got ({ns[i]}, 2) and ({nm[i]}, 3)")
    offs = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(ns, out=offs[1:])
    return offs, sensor, model


def packModelPoints(allModelPoints):
    M = len(allModelPoints)
    nm, model = _stack(list(allModelPoints), 3)
    offs = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(nm, out=offs[1:])
    return offs, model


class RefineEngine:
    def __init__(self, model, dtype="f64", device=0):
        self._lib = nat.loadLibrary()
        nat.requireDevice()
        self.device = int(device)
        self.modelId = MODEL_IDS[model] if isinstance(model, str) else int(model)
        self.dtypeId = DTYPE_IDS[dtype] if isinstance(dtype, str) else int(dtype)
        self.L = NUM_SHARED[self.modelId]
        self.C = self.L + 6
        h = ctypes.c_void_p()
        nat.check(self._lib.calib_create(self.modelId, self.dtypeId, int(device), ctypes.byref(h)))
        self._h = h
        self.M = 0
        self.MN = 0
        self._lmMaxIters = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.calib_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def K(self):
        return self.L + 6 * self.M

    # ---- problem -------------------------------------------------------------------------
    def setProblem(self, viewOffsets, sensorPoints, modelPoints):
        offs = np.ascontiguousarray(viewOffsets, dtype=np.int64)
        if offs.ndim != 1 or offs.shape[0] < 1:
            raise ValueError("viewOffsets must be a 1-D array of length M+1")
        MN = int(offs[-1])
        model = np.ascontiguousarray(modelPoints, dtype=np.float64).reshape(-1, 3)
        if model.shape[0] != MN:
            raise ValueError(f"Expected shape ({MN}, 3), got {model.shape}")
        sensor = None
        if sensorPoints is not None:
            sensor = np.ascontiguousarray(sensorPoints, dtype=np.float64).reshape(-1, 2)
            if sensor.shape[0] != MN:
                raise ValueError(f"Expected shape ({MN}, 2), got {sensor.shape}")
        nat.check(self._lib.calib_set_problem(self._h, offs.shape[0] - 1, nat.i64ptr(offs),
                                              nat.dptr(sensor), nat.dptr(model)))
        self.M = offs.shape[0] - 1
        self.MN = MN
        self.viewOffsets = offs

    def setProblemViews(self, viewOffsets, sensorAddresses, modelAddresses):
        """The same upload from per-view arrays (viewPointers): nothing is stacked on the host, the library's staged
        upload gathers from the views (calib_set_problem_views). The caller keeps the arrays alive during the call."""
        offs = np.ascontiguousarray(viewOffsets, dtype=np.int64)
        M = offs.shape[0] - 1
        sa = None if sensorAddresses is None else np.ascontiguousarray(sensorAddresses, dtype=np.uint64)
        ma = np.ascontiguousarray(modelAddresses, dtype=np.uint64)
        if ma.shape[0] != M or (sa is not None and sa.shape[0] != M):
            raise ValueError(f"Expected {M} view addresses")
        nat.check(self._lib.calib_set_problem_views(self._h, M, nat.i64ptr(offs),
                                                    None if sa is None else ctypes.c_void_p(sa.ctypes.data),
                                                    ctypes.c_void_p(ma.ctypes.data)))
        self.M = M
        self.MN = int(offs[-1])
        self.viewOffsets = offs

    def setStream(self, hipStream):
        """Run on an existing HIP stream, given as an integer handle (0 = the default stream, which
        is torch's current stream unless changed); None = back to the engine's own stream."""
        if hipStream is None:
            nat.check(self._lib.calib_set_stream(self._h, None, 1))
        else:
            nat.check(self._lib.calib_set_stream(self._h, ctypes.c_void_p(int(hipStream)), 0))

    def setLmMode(self, mode):
        """'fused' (default) or 'two_kernel' (materialise the compact J in HBM), see calib_lm.h."""
        ids = {"fused": nat.LM_FUSED, "two_kernel": nat.LM_TWO_KERNEL}
        nat.check(self._lib.calib_set_lm_mode(self._h, ids[mode] if isinstance(mode, str) else int(mode)))

    def fusedForm(self):
        """-> (share, waves): share > 0 when the loaded problem's fused rounds run in the stream form (4-point groups
        per wave, waves of the launch), (0, 0) for one view item per wave (calib_fused_form)"""
        share, waves = ctypes.c_int(0), ctypes.c_int(0)
        nat.check(self._lib.calib_fused_form(self._h, ctypes.byref(share), ctypes.byref(waves)))
        return share.value, waves.value

    def _P(self, P):
        P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).ravel())
        if P.shape[0] != self.K:
            raise ValueError(f"Expected shape ({self.K},), got {P.shape}")
        return P

    # ---- single evaluations --------------------------------------------------------------
    def evaluate(self, P, wantY=False, wantR=False, wantJ=False):
        """-> dict with any of y (MN,2), r (MN,2), Jc (MN,2,C) and sse."""
        P = self._P(P)
        y = np.empty((self.MN, 2)) if wantY else None
        r = np.empty((self.MN, 2)) if wantR else None
        Jc = np.empty((self.MN, 2, self.C)) if wantJ else None
        sse = ctypes.c_double(0.0)
        nat.check(self._lib.calib_eval(self._h, nat.dptr(P), nat.dptr(y), nat.dptr(r), nat.dptr(Jc),
                                       ctypes.byref(sse)))
        return {"y": y, "r": r, "Jc": Jc, "sse": sse.value}

    def normalEquations(self, P):
        """-> B (L,L), E (M,L,6), V (M,6,6), g (K,): the block-arrow J^T J and J^T r."""
        P = self._P(P)
        B = np.empty((self.L, self.L))
        E = np.empty((self.M, self.L, 6))
        V = np.empty((self.M, 6, 6))
        g = np.empty(self.K)
        nat.check(self._lib.calib_normal_eq(self._h, nat.dptr(P), nat.dptr(B), nat.dptr(E), nat.dptr(V),
                                            nat.dptr(g)))
        return B, E, V, g

    def stepDelta(self, P, lam):
        P = self._P(P)
        d = np.empty(self.K)
        nat.check(self._lib.calib_lm_step_delta(self._h, nat.dptr(P), float(lam), nat.dptr(d)))
        return d

    # ---- LM ------------------------------------------------------------------------------
    def refine(self, P0, maxIters, lamInit=LAMBDA_INITIAL, lamMin=LAMBDA_MIN, lamMax=LAMBDA_MAX,
               errMin=PT_ERROR_MIN):
        """Whole loop of src/calibrate.py:143-171 on the device.
        -> (sse [pre-update, as the reference returns], P, iters, trace (iters, 5+L))."""
        P = self._P(P0).copy()
        if int(maxIters) <= 0:
            raise UnboundLocalError("local variable 'Pt_error' referenced before assignment "
                                    "(maxIters=0, src/calibrate.py:171)")
        trace = np.zeros((int(maxIters), nat.TRACE_HEADER + self.L))
        sse = ctypes.c_double(0.0)
        iters = ctypes.c_int(0)
        nat.check(self._lib.calib_refine(self._h, nat.dptr(P), int(maxIters), float(lamInit), float(lamMin),
                                         float(lamMax), float(errMin), ctypes.byref(sse),
                                         ctypes.byref(iters), nat.dptr(trace)))
        return sse.value, P, iters.value, trace[:iters.value]

    def refineAWk(self, A, W, k, maxIters, lamInit=LAMBDA_INITIAL, lamMin=LAMBDA_MIN, lamMax=LAMBDA_MAX,
                  errMin=PT_ERROR_MIN):
        """(A, W, k) in, refined (A, W, k) out: compose -> LM loop -> decompose, all behind the C-ABI
        (calib_refine_awk). -> (sse, A (3,3), W (M,4,4), k, iters, trace)"""
        if int(maxIters) <= 0:
            raise UnboundLocalError("local variable 'Pt_error' referenced before assignment "
                                    "(maxIters=0, src/calibrate.py:171)")
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3).copy()
        W = np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(-1, 4, 4)).copy()
        k = np.ascontiguousarray(k, dtype=np.float64).ravel().copy()
        if W.shape[0] != self.M or k.shape[0] != self.L - 5:
            raise ValueError(f"Expected {self.M} poses and {self.L - 5} distortion coefficients, "
                             f"got {W.shape[0]} and {k.shape[0]}")
        trace = np.zeros((int(maxIters), nat.TRACE_HEADER + self.L))
        sse = ctypes.c_double(0.0)
        iters = ctypes.c_int(0)
        nat.check(self._lib.calib_refine_awk(self._h, nat.dptr(A), nat.dptr(W), nat.dptr(k), int(maxIters),
                                             float(lamInit), float(lamMin), float(lamMax), float(errMin),
                                             ctypes.byref(sse), ctypes.byref(iters), nat.dptr(trace)))
        return sse.value, A, W, k, iters.value, trace[:iters.value]

    # stepping form (multi-GPU shards, benchmarks)
    def lmBegin(self, P0, maxIters, lamInit=LAMBDA_INITIAL, lamMin=LAMBDA_MIN, lamMax=LAMBDA_MAX,
                errMin=PT_ERROR_MIN):
        P = self._P(P0)
        self._lmMaxIters = int(maxIters)
        nat.check(self._lib.calib_lm_begin(self._h, nat.dptr(P), int(maxIters), float(lamInit),
                                           float(lamMin), float(lamMax), float(errMin)))

    def reduceSize(self):
        n = ctypes.c_int64(0)
        nat.check(self._lib.calib_lm_reduce_size(self._h, ctypes.byref(n)))
        return n.value

    def bindReduceBuffer(self, devicePointer):
        nat.check(self._lib.calib_lm_bind_reduce_buffer(self._h, ctypes.c_void_p(devicePointer or 0)))

    def lmLocal(self):
        nat.check(self._lib.calib_lm_local(self._h))

    def lmUpdate(self):
        nat.check(self._lib.calib_lm_update(self._h))

    def lmRun(self, rounds, checkEvery=0):
        nat.check(self._lib.calib_lm_run(self._h, int(rounds), int(checkEvery)))

    def lmRunSharded(self, rounds, checkEvery=0):
        """whole rounds with the in-library all-reduce between local and update (every rank calls it)"""
        nat.check(self._lib.calib_lm_run_sharded(self._h, int(rounds), int(checkEvery)))

    # ---- peer exchange over xGMI (include/calib_lm.h: calib_peer_*) ------------------------------
    def peerPrepare(self, nranks, rank):
        """-> this rank's 64-byte IPC handle of its slot memory (bytes)."""
        buf = (ctypes.c_ubyte * 64)()
        nat.check(self._lib.calib_peer_prepare(self._h, int(nranks), int(rank), ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def peerConnect(self, handles, timeoutSeconds=60.0):
        """handles: the ranks' peerPrepare results in rank order."""
        blob = b"".join(handles)
        buf = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
        nat.check(self._lib.calib_peer_connect(self._h, ctypes.cast(buf, ctypes.c_void_p), float(timeoutSeconds)))

    def peerSelfTest(self, rounds=64, timeoutSeconds=10.0):
        nat.check(self._lib.calib_peer_selftest(self._h, int(rounds), float(timeoutSeconds)))

    def peerShutdown(self):
        nat.check(self._lib.calib_peer_shutdown(self._h))

    # ---- in-library all-reduce (include/calib_lm.h: calib_rccl_*) -------------------------------
    def rcclLoad(self, librcclPath):
        nat.check(self._lib.calib_rccl_load(str(librcclPath).encode()))

    def rcclUniqueId(self):
        buf = ctypes.create_string_buffer(128)
        nat.check(self._lib.calib_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf.raw)

    def rcclInit(self, nranks, rank, uniqueId, timeoutSeconds=120.0):
        buf = ctypes.create_string_buffer(bytes(uniqueId), 128)
        nat.check(self._lib.calib_rccl_init_deadline(self._h, int(nranks), int(rank), ctypes.cast(buf, ctypes.c_void_p),
                                                     float(timeoutSeconds)))

    def rcclSelfTest(self, timeoutSeconds=30.0):
        nat.check(self._lib.calib_rccl_selftest(self._h, float(timeoutSeconds)))

    def rcclShutdown(self):
        nat.check(self._lib.calib_rccl_shutdown(self._h))

    def lmAllReduce(self):
        nat.check(self._lib.calib_lm_allreduce(self._h))

    def synchronize(self):
        """wait for everything enqueued on the engine's stream (calib_synchronize)"""
        nat.check(self._lib.calib_synchronize(self._h))

    def lmDone(self):
        d = ctypes.c_int(0)
        nat.check(self._lib.calib_lm_done(self._h, ctypes.byref(d)))
        return bool(d.value)

    def peekTrace(self, it):
        """-> (row (5+L,), itersExecuted) of the running loop; synchronises."""
        row = np.zeros(nat.TRACE_HEADER + self.L)
        n = ctypes.c_int(0)
        nat.check(self._lib.calib_lm_peek_trace(self._h, int(it), nat.dptr(row), ctypes.byref(n)))
        return row, n.value

    def lmEnd(self):
        P = np.empty(self.K)
        trace = np.zeros((max(self._lmMaxIters, 1), nat.TRACE_HEADER + self.L))
        sse = ctypes.c_double(0.0)
        iters = ctypes.c_int(0)
        nat.check(self._lib.calib_lm_end(self._h, nat.dptr(P), ctypes.byref(sse), ctypes.byref(iters),
                                         nat.dptr(trace)))
        return sse.value, P, iters.value, trace[:iters.value]

    # ---- profiling -----------------------------------------------------------------------
    def profileEnable(self, on=True, every=1):
        """HIP-event timing of the dominant kernels; every = N times only every N-th launch of each
        kernel (timing all of them slows the loop being measured)."""
        nat.check(self._lib.calib_profile_enable(self._h, max(1, int(every)) if on else 0))

    def profileRead(self, which):
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        nat.check(self._lib.calib_profile_read(self._h, int(which), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


# Up to this many bytes of correspondences ResidentProblem.get() decides by CONTENT whether the engine already holds
# the problem (compare against private copies); above it, comparing costs more than uploading again -- a host
# compare runs at a few GB/s on one core, calib_set_problem at 21-33 GB/s plus ~0.3 ms -- so large problems are
# simply uploaded, unless the caller vouches for them (sameProblem=True).
RESIDENT_COMPARE_LIMIT = 4 << 20


class ResidentProblem:
    """One RefineEngine kept alive with its correspondences resident in HBM, reused while the caller keeps
    asking about the same problem (same offsets and model points; sensor points when they are needed):
    Calibrator.projectAllPoints / _computeReprojectionError / refineCalibrationParameters and
    ProjectionJacobian.compute then pay engine creation and the upload once, not per call.
    Small problems (<= RESIDENT_COMPARE_LIMIT bytes) are recognised by content; a large one is uploaded on every
    call -- cheaper than comparing it, and never stale -- unless the caller passes sameProblem=True ("these are the
    arrays of my previous call, unchanged"), which skips both the compare and the upload.
    lastSeconds: host time the last get() spent in {"compare", "upload"}."""

    def __init__(self, modelId, dtype, device):
        self.modelId, self.dtype, self.device = modelId, dtype, device
        self.eng = None
        self._offs = self._sensor = self._model = None
        self._shape = None          # (M, MN, has sensor points) of what the engine holds
        self.uploads = 0
        self.lastSeconds = {"compare": 0.0, "upload": 0.0}

    def get(self, viewOffsets, sensorPoints, modelPoints, sameProblem=False):
        import time
        offs = np.ascontiguousarray(viewOffsets, dtype=np.int64)
        shape = (offs.shape[0] - 1, int(offs[-1]) if offs.shape[0] else 0)
        self.lastSeconds = {"compare": 0.0, "upload": 0.0}
        if sameProblem and self.eng is not None and self._shape is not None and self._shape[:2] == shape \
                and (sensorPoints is None or self._shape[2]):
            return self.eng
        nbytes = shape[1] * (24 + (16 if sensorPoints is not None else 0))
        small = nbytes <= RESIDENT_COMPARE_LIMIT
        t0 = time.perf_counter()
        same = (small and self.eng is not None and self._offs is not None and np.array_equal(offs, self._offs)
                and np.array_equal(modelPoints, self._model)
                and (sensorPoints is None or (self._sensor is not None and np.array_equal(sensorPoints, self._sensor))))
        self.lastSeconds["compare"] = time.perf_counter() - t0
        if not same:
            if self.eng is None:
                self.eng = RefineEngine(self.modelId, self.dtype, self.device)
            # the keys are PRIVATE copies (a caller that changes its arrays in place must not be compared with
            # itself), and they describe the engine only once the upload has succeeded
            self._offs = self._sensor = self._model = self._shape = None
            t0 = time.perf_counter()
            self.eng.setProblem(offs, sensorPoints, modelPoints)
            self.lastSeconds["upload"] = time.perf_counter() - t0
            if small:
                self._offs = offs.copy()
                self._sensor = None if sensorPoints is None else np.array(sensorPoints, dtype=np.float64, copy=True)
                self._model = np.array(modelPoints, dtype=np.float64, copy=True)
            self._shape = shape + (sensorPoints is not None,)
            self.uploads += 1
        return self.eng

    def getFromDetections(self, allDetections):
        """get() for the reference's own argument, a list of per-view (sensor, model) arrays: a large problem is
        uploaded straight from the views (viewPointers + calib_set_problem_views: no stacked copy on the host); small
        ones, and lists the helper cannot take, are stacked (packDetections) and go through get().
        lastSeconds gains "pack": the host time spent stacking / collecting the view addresses."""
        import time
        t0 = time.perf_counter()
        vp = viewPointers(allDetections)
        if vp is None or int(vp[0][-1]) * 40 <= RESIDENT_COMPARE_LIMIT:
            offs, sensor, model = packDetections(allDetections)
            tPack = time.perf_counter() - t0
            eng = self.get(offs, sensor, model)
            self.lastSeconds["pack"] = tPack
            return eng
        offs, sAddr, mAddr = vp
        tPack = time.perf_counter() - t0
        if self.eng is None:
            self.eng = RefineEngine(self.modelId, self.dtype, self.device)
        self._offs = self._sensor = self._model = self._shape = None
        t0 = time.perf_counter()
        self.eng.setProblemViews(offs, sAddr, mAddr)
        self.lastSeconds = {"compare": 0.0, "upload": time.perf_counter() - t0, "pack": tPack}
        self._shape = (offs.shape[0] - 1, int(offs[-1]), True)
        self.uploads += 1
        return self.eng

    def close(self):
        if self.eng is not None:
            self.eng.close()
            self.eng = None
        self._offs = self._sensor = self._model = self._shape = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def refineHomographies(Hs, viewOffsets, sensorPoints, modelPoints, maxIters=20, device=0):
    """LM polish of every view's homography on the device (src/calibrate.py:60-111).
    Hs (M,3,3) -> (M,3,3) with H[2,2] = 1."""
    H = np.ascontiguousarray(np.asarray(Hs, dtype=np.float64).reshape(-1, 3, 3)).copy()
    offs = np.ascontiguousarray(viewOffsets, dtype=np.int64)
    s = np.ascontiguousarray(sensorPoints, dtype=np.float64).reshape(-1, 2)
    m = np.ascontiguousarray(modelPoints, dtype=np.float64).reshape(-1, 3)
    if offs.shape[0] - 1 != H.shape[0] or s.shape[0] != offs[-1] or m.shape[0] != offs[-1]:
        raise ValueError(f"Expected {offs.shape[0] - 1} homographies and {int(offs[-1])} points")
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_refine_homographies(H.shape[0], nat.i64ptr(offs), nat.dptr(s), nat.dptr(m),
                                                          nat.dptr(H), int(maxIters), int(device)))
    return H


def composeParameters(modelId, A, W, k, device=0):
    """(A (3,3), W (M,4,4), k) -> P (L + 6M,): Euler angles (degrees) of every pose on the device
    (src/calibrate.py:199-229, src/mathutils.py:13-33)."""
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3)
    W = np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(-1, 4, 4))
    k = np.ascontiguousarray(k, dtype=np.float64).ravel()
    if k.shape[0] != NUM_SHARED[modelId] - 5:
        raise ValueError(f"Expected {NUM_SHARED[modelId] - 5} distortion coefficients, got {k.shape[0]}")
    P = np.empty(NUM_SHARED[modelId] + 6 * W.shape[0])
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_compose_params(modelId, W.shape[0], nat.dptr(A), nat.dptr(W), nat.dptr(k),
                                                     nat.dptr(P), int(device)))
    return P


def decomposeParameters(modelId, P, device=0):
    """P (L + 6M,) -> (A (3,3), W (M,4,4), k): src/calibrate.py:231-267 with the poses rebuilt on the device."""
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).ravel())
    L = NUM_SHARED[modelId]
    if P.shape[0] < L or (P.shape[0] - L) % 6:
        raise ValueError(f"Expected shape ({L} + 6 M,), got {P.shape}")
    M = (P.shape[0] - L) // 6
    A, W, k = np.empty((3, 3)), np.empty((M, 4, 4)), np.empty(L - 5)
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_decompose_params(modelId, M, nat.dptr(P), nat.dptr(A), nat.dptr(W), nat.dptr(k),
                                                       int(device)))
    return A, W, k


def homographyJacobian(h, modelPoints, device=0):
    """(2N, 9) Jacobian of the homography projection wrt h (src/jacobian.py:88-121), on the device."""
    h = np.ascontiguousarray(np.asarray(h, dtype=np.float64).ravel())
    m = np.ascontiguousarray(modelPoints, dtype=np.float64)
    if h.shape[0] != 9:
        raise ValueError(f"Expected shape (9,), got {h.shape}")
    if m.ndim != 2 or m.shape[1] != 3:
        raise ValueError(f"Expected shape (None, 3), got {m.shape}")
    J = np.empty((2 * m.shape[0], 9))
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_homography_jacobian(m.shape[0], nat.dptr(h), nat.dptr(m), nat.dptr(J), int(device)))
    return J


def _packedViews(viewOffsets, sensorPoints, modelPoints):
    offs = np.ascontiguousarray(viewOffsets, dtype=np.int64)
    s = np.ascontiguousarray(sensorPoints, dtype=np.float64).reshape(-1, 2)
    m = np.ascontiguousarray(modelPoints, dtype=np.float64).reshape(-1, 3)
    if s.shape[0] != offs[-1] or m.shape[0] != offs[-1]:
        raise ValueError(f"Expected {int(offs[-1])} points, got {s.shape[0]} and {m.shape[0]}")
    return offs, s, m


def estimateHomographies(viewOffsets, sensorPoints, modelPoints, refineIters=20, device=0):
    """Normalised DLT + LM polish of every view's homography on the device
    (src/linearcalibrate.py:7-58, src/calibrate.py:60-111). -> (M,3,3), H[2,2] = 1."""
    offs, s, m = _packedViews(viewOffsets, sensorPoints, modelPoints)
    H = np.empty((offs.shape[0] - 1, 3, 3))
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_estimate_homographies(H.shape[0], nat.i64ptr(offs), nat.dptr(s), nat.dptr(m),
                                                            nat.dptr(H), int(refineIters), int(device)))
    return H


def computeExtrinsics(Hs, A, device=0):
    """World-to-camera poses (M,4,4) from homographies and A (src/linearcalibrate.py:306-371)."""
    H = np.ascontiguousarray(np.asarray(Hs, dtype=np.float64).reshape(-1, 3, 3))
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3)
    W = np.empty((H.shape[0], 4, 4))
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_compute_extrinsics(H.shape[0], nat.dptr(A), nat.dptr(H), nat.dptr(W), int(device)))
    return W


def distortionNormalEquations(modelId, viewOffsets, sensorPoints, modelPoints, A, W, device=0):
    """D^T D and D^T Ddot of the linear distortion estimate (src/distortion.py:110-191, 222-271)."""
    offs, s, m = _packedViews(viewOffsets, sensorPoints, modelPoints)
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3)
    W = np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(-1, 4, 4))
    n = 5 if modelId == nat.MODEL_RADTAN else 4
    G, g = np.empty((n, n)), np.empty(n)
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_distortion_normal_equations(modelId, offs.shape[0] - 1, nat.i64ptr(offs),
                                                                  nat.dptr(s), nat.dptr(m), nat.dptr(A), nat.dptr(W),
                                                                  nat.dptr(G), nat.dptr(g), int(device)))
    return G, g


def distortPoints(modelId, x, k):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
    k = np.ascontiguousarray(k, dtype=np.float64).ravel()
    out = np.empty_like(x)
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_distort_points(modelId, x.shape[0], nat.dptr(x), nat.dptr(k),
                                                     nat.dptr(out)))
    return out


def projectWithDistortion(modelId, A, X, k):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3)
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
    k = np.ascontiguousarray(k, dtype=np.float64).ravel()
    out = np.empty((X.shape[0], 2))
    nat.requireDevice()
    nat.check(nat.loadLibrary().calib_project_with_distortion(modelId, X.shape[0], nat.dptr(A), nat.dptr(X),
                                                              nat.dptr(k), nat.dptr(out)))
    return out
