/* _fastpack: the one place where the Python host side needs the CPython C API.
 *
 * The reference's callers hold their correspondences as a LIST of per-view arrays -- allDetections =
 * [(sensorPoints (N_i,2), modelPoints (N_i,3)), ...] (src/calibrate.py:117-118) -- and stack them with np.vstack per
 * call (getSensorPoints, src/calibrate.py:277-282). For 10 000 views that stacking is the largest single cost of a
 * refineCalibrationParameters call here (12 ms of 22, the LM loop itself 2.6 ms): a fresh 80 MB array is page-faulted
 * in and filled by one thread, only to be copied once more into the pinned staging of the upload.
 * view_pointers() walks the list once and hands back, per view, the row count and the address of the view's (already
 * C-contiguous float64) data; calib_set_problem_views (include/calib_lm.h) then gathers straight from those
 * addresses into its pinned staging buffers on its four upload threads -- nothing is stacked on the host.
 *
 * Plain C against Python.h + numpy/arrayobject.h, built by csrc/Makefile into ../lib/. Optional: when the module
 * is missing or a view is not a C-contiguous float64 (N, width) ndarray, engine.py stacks with numpy as before. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

/* view_pointers(detections, which, width) -> (counts int64[M], addresses uint64[M]) | None
 *   detections: a list / tuple of views; which < 0: every view IS the array; which >= 0: a view is a pair and
 *   element `which` of it is the array (0 = sensor points, 1 = model points).
 * None when any view is not eligible (the caller falls back to numpy). The addresses stay valid for as long as the
 * caller keeps the list (and so the arrays) alive. */
static PyObject* view_pointers(PyObject* self, PyObject* args) {
    PyObject* seq;
    int which, width;
    (void)self;
    if (!PyArg_ParseTuple(args, "Oii", &seq, &which, &width)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "detections must be a sequence");
    if (!fast) return NULL;
    const Py_ssize_t M = PySequence_Fast_GET_SIZE(fast);
    npy_intp dims[1] = {(npy_intp)M};
    PyObject* counts = PyArray_SimpleNew(1, dims, NPY_INT64);
    PyObject* addrs = PyArray_SimpleNew(1, dims, NPY_UINT64);
    if (!counts || !addrs) { Py_XDECREF(counts); Py_XDECREF(addrs); Py_DECREF(fast); return NULL; }
    npy_int64* pc = (npy_int64*)PyArray_DATA((PyArrayObject*)counts);
    npy_uint64* pa = (npy_uint64*)PyArray_DATA((PyArrayObject*)addrs);
    int ok = 1;
    for (Py_ssize_t i = 0; i < M && ok; ++i) {
        PyObject* o = PySequence_Fast_GET_ITEM(fast, i);                    /* borrowed */
        if (which >= 0) {
            if ((PyTuple_Check(o) || PyList_Check(o)) && PySequence_Fast_GET_SIZE(o) == 2)
                o = PySequence_Fast_GET_ITEM(o, which);                     /* borrowed */
            else { ok = 0; break; }
        }
        if (!PyArray_CheckExact(o)) { ok = 0; break; }
        PyArrayObject* a = (PyArrayObject*)o;
        if (PyArray_TYPE(a) != NPY_DOUBLE || PyArray_NDIM(a) != 2 || PyArray_DIM(a, 1) != width ||
            !PyArray_IS_C_CONTIGUOUS(a) || !PyArray_ISALIGNED(a) || PyArray_ISBYTESWAPPED(a)) { ok = 0; break; }
        pc[i] = (npy_int64)PyArray_DIM(a, 0);
        pa[i] = (npy_uint64)(uintptr_t)PyArray_DATA(a);
    }
    Py_DECREF(fast);
    if (!ok) { Py_DECREF(counts); Py_DECREF(addrs); Py_RETURN_NONE; }
    PyObject* out = PyTuple_Pack(2, counts, addrs);
    Py_DECREF(counts);
    Py_DECREF(addrs);
    return out;
}

static PyMethodDef methods[] = {
    {"view_pointers", view_pointers, METH_VARARGS,
     "view_pointers(detections, which, width) -> (counts int64[M], addresses uint64[M]) or None"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastpack",
                                       "per-view row counts and data addresses of a list of float64 arrays", -1, methods,
                                       NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastpack(void) {
    import_array();
    return PyModule_Create(&moduledef);
}
