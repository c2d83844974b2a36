/* CPython glue for the per-iteration call of the hot path (host plumbing only).
 *
 * ctypes spends 1-2 us converting the ten arguments of raftcorr_lookup_fwd, as much as the launch
 * itself; a RAFT pair makes 12-32 such calls.  This module calls the SAME C-ABI entry point
 * (include/raftcorr.h) through its address, with arguments parsed by the C API.  raft_amd/corr.py
 * uses it when it is built and the ctypes handle otherwise -- the kernels are identical.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*lookup_fwd_fn)(const float* const*, int, int, int, int, int, const float*, int, float*, void*);

/* lookup_fwd(fn, levels, B, H, W, L, radius, coords, mode, out, stream) -> status */
static PyObject* lookup_fwd(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 11) {
        PyErr_SetString(PyExc_TypeError, "lookup_fwd expects 11 arguments");
        return NULL;
    }
    uintptr_t p[4];
    const int pidx[4] = {0, 1, 7, 9};
    for (int i = 0; i < 4; ++i) {
        p[i] = (uintptr_t)PyLong_AsUnsignedLongLong(args[pidx[i]]);
        if (p[i] == (uintptr_t)-1 && PyErr_Occurred()) return NULL;
    }
    uintptr_t stream = 0;
    if (args[10] != Py_None) {
        stream = (uintptr_t)PyLong_AsUnsignedLongLong(args[10]);
        if (stream == (uintptr_t)-1 && PyErr_Occurred()) return NULL;
    }
    long v[6];
    const int vidx[6] = {2, 3, 4, 5, 6, 8};
    for (int i = 0; i < 6; ++i) {
        v[i] = PyLong_AsLong(args[vidx[i]]);
        if (v[i] == -1 && PyErr_Occurred()) return NULL;
    }
    if (!p[0]) {
        PyErr_SetString(PyExc_ValueError, "lookup_fwd: null function address");
        return NULL;
    }
    const int rc = ((lookup_fwd_fn)p[0])((const float* const*)p[1], (int)v[0], (int)v[1], (int)v[2], (int)v[3],
                                         (int)v[4], (const float*)p[2], (int)v[5], (float*)p[3], (void*)stream);
    return PyLong_FromLong(rc);
}

typedef int (*lookup_fwd_vol_fn)(const void* const*, int, int, int, int, int, int, const float*, int, float*, void*);

/* lookup_fwd_vol(fn, levels, volume_dtype, B, H, W, L, radius, coords, mode, out, stream) -> status */
static PyObject* lookup_fwd_vol(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 12) {
        PyErr_SetString(PyExc_TypeError, "lookup_fwd_vol expects 12 arguments");
        return NULL;
    }
    uintptr_t p[4];
    const int pidx[4] = {0, 1, 8, 10};
    for (int i = 0; i < 4; ++i) {
        p[i] = (uintptr_t)PyLong_AsUnsignedLongLong(args[pidx[i]]);
        if (p[i] == (uintptr_t)-1 && PyErr_Occurred()) return NULL;
    }
    uintptr_t stream = 0;
    if (args[11] != Py_None) {
        stream = (uintptr_t)PyLong_AsUnsignedLongLong(args[11]);
        if (stream == (uintptr_t)-1 && PyErr_Occurred()) return NULL;
    }
    long v[7];
    const int vidx[7] = {2, 3, 4, 5, 6, 7, 9};
    for (int i = 0; i < 7; ++i) {
        v[i] = PyLong_AsLong(args[vidx[i]]);
        if (v[i] == -1 && PyErr_Occurred()) return NULL;
    }
    if (!p[0]) {
        PyErr_SetString(PyExc_ValueError, "lookup_fwd_vol: null function address");
        return NULL;
    }
    const int rc = ((lookup_fwd_vol_fn)p[0])((const void* const*)p[1], (int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)v[4],
                                             (int)v[5], (const float*)p[2], (int)v[6], (float*)p[3], (void*)stream);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"lookup_fwd_vol", (PyCFunction)(void (*)(void))lookup_fwd_vol, METH_FASTCALL,
     "lookup_fwd_vol(fn, levels, volume_dtype, B, H, W, L, radius, coords, mode, out, stream) -> int"},
    {"lookup_fwd", (PyCFunction)(void (*)(void))lookup_fwd, METH_FASTCALL,
     "lookup_fwd(fn, levels, B, H, W, L, radius, coords, mode, out, stream) -> int"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_pyfast", "fast call path of raft_amd", -1, methods,
                                       NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pyfast(void) { return PyModule_Create(&moduledef); }
