/* pyfast.c -- PyAscore.score()'s way into pya_score_one without ctypes: the reference's score() is a compiled
 * (Cython) method (Ascore.pyx:103-152), and a ctypes call with thirteen converted arguments plus five
 * `ndarray.ctypes.data` lookups costs more than the launch it leads to.  One function: takes the arguments as
 * score() received them, checks what the fast way needs (exact argument types and layouts -- anything else is
 * answered with None and goes the checked Python way, which raises what the reference raises), calls
 * pya_score_one through the function pointer it was given (no link-time dependency on the library), and
 * returns the results as Python scalars and bytes.
 *
 *   gcc -O2 -fPIC -shared -I<python include> pyfast.c -o ../_fast.so        (pyascore_amd/build.py does it) */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* include/pyascore_hip.h: pya_results and pya_score_one */
typedef struct {
    uint32_t max_k;
    float *best_score;
    uint64_t *best_sig;
    int32_t *n_sig;
    float *ascores;
    uint64_t *alt_mask;
} fast_results;
typedef int (*score_one_fn)(void *h, const double *mz, const double *inten, uint64_t n_peaks, const uint8_t *pep,
                            uint64_t L, int32_t n_of_mod, int32_t max_charge, const uint32_t *aux_pos,
                            const float *aux_mass, uint64_t n_aux, uint32_t flags, const fast_results *out);

static PyTypeObject *g_ndarray = NULL;

/* a 1-D C-contiguous buffer of an exact ndarray with the given struct format and item size */
static int get_vec(PyObject *o, Py_buffer *v, const char *fmt, Py_ssize_t item) {
    if (!g_ndarray || !PyObject_TypeCheck(o, g_ndarray)) return 0;
    if (PyObject_GetBuffer(o, v, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) != 0) {
        PyErr_Clear();
        return 0;
    }
    const char *f = v->format ? v->format : "B";
    if (*f == '@' || *f == '=' || *f == '<') f++;              /* (native / little endian prefixes) */
    if (v->ndim != 1 || v->itemsize != item || strcmp(f, fmt) != 0) {
        PyBuffer_Release(v);
        return 0;
    }
    return 1;
}

/* setup(ndarray_type) */
static PyObject *fast_setup(PyObject *self, PyObject *arg) {
    if (!PyType_Check(arg)) {
        PyErr_SetString(PyExc_TypeError, "setup(numpy.ndarray)");
        return NULL;
    }
    Py_XDECREF((PyObject *)g_ndarray);
    Py_INCREF(arg);
    g_ndarray = (PyTypeObject *)arg;
    Py_RETURN_NONE;
}

/* score_one(fn, handle, mz, inten, peptide, n_of_mod, max_charge, aux_pos, aux_mass)
 *   -> None (not for the fast way) | (rc,) | (0, best_score, best_sig, n_sig, ascores bytes, alt_mask bytes, n_of_mod) */
static PyObject *fast_score_one(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 9) {
        PyErr_SetString(PyExc_TypeError, "score_one takes 9 arguments");
        return NULL;
    }
    score_one_fn fn = (score_one_fn)PyLong_AsVoidPtr(args[0]);
    void *h = PyLong_AsVoidPtr(args[1]);
    if (PyErr_Occurred()) return NULL;
    if (!fn || !h) Py_RETURN_NONE;
    if (!PyUnicode_CheckExact(args[4])) Py_RETURN_NONE;
    /* (Python and numpy integers alike, as a Cython size_t argument takes them) */
    long nz[2];
    for (int i = 0; i < 2; i++) {
        PyObject *ix = PyIndex_Check(args[5 + i]) ? PyNumber_Index(args[5 + i]) : NULL;
        if (!ix) {
            PyErr_Clear();
            Py_RETURN_NONE;
        }
        int overflow = 0;
        nz[i] = PyLong_AsLongAndOverflow(ix, &overflow);
        Py_DECREF(ix);
        if (overflow || nz[i] < 0 || nz[i] > 1000) Py_RETURN_NONE;
    }
    const long n_of_mod = nz[0], z = nz[1];
    if (n_of_mod > 64) Py_RETURN_NONE;
    Py_ssize_t L = 0;
    const char *pep = PyUnicode_AsUTF8AndSize(args[4], &L);
    if (!pep) {
        PyErr_Clear();
        Py_RETURN_NONE;
    }
    Py_buffer mz, it, ap, am;
    int have_aux = 0;
    if (!get_vec(args[2], &mz, "d", 8)) Py_RETURN_NONE;
    if (!get_vec(args[3], &it, "d", 8)) {
        PyBuffer_Release(&mz);
        Py_RETURN_NONE;
    }
    PyObject *ret = NULL;
    if (mz.len != it.len) goto slow;
    if (args[7] != Py_None && args[8] != Py_None) {
        if (!get_vec(args[7], &ap, "I", 4)) goto slow;
        if (!get_vec(args[8], &am, "f", 4)) {
            PyBuffer_Release(&ap);
            goto slow;
        }
        have_aux = 1;
        if (ap.len != am.len) goto slow_aux;
    }
    {
        float best_score = 0.f, asc[64];
        uint64_t best_sig = 0, alt[64];
        int32_t n_sig = 0;
        const uint32_t k = n_of_mod > 1 ? (uint32_t)n_of_mod : 1u;
        memset(asc, 0, sizeof asc);
        memset(alt, 0, sizeof alt);
        fast_results r = {k, &best_score, &best_sig, &n_sig, asc, alt};
        int rc;
        Py_BEGIN_ALLOW_THREADS
        rc = fn(h, (const double *)mz.buf, (const double *)it.buf, (uint64_t)(mz.len / 8), (const uint8_t *)pep, (uint64_t)L,
                (int32_t)n_of_mod, (int32_t)z, have_aux ? (const uint32_t *)ap.buf : NULL, have_aux ? (const float *)am.buf : NULL,
                have_aux ? (uint64_t)(ap.len / 4) : 0u, 0u, &r);
        Py_END_ALLOW_THREADS
        if (rc != 0) ret = Py_BuildValue("(i)", rc);
        else
            ret = Py_BuildValue("(idKiy#y#i)", 0, (double)best_score, (unsigned long long)best_sig, (int)n_sig, (const char *)asc,
                                (Py_ssize_t)(k * 4), (const char *)alt, (Py_ssize_t)(k * 8), (int)n_of_mod);
    }
    if (have_aux) {
        PyBuffer_Release(&ap);
        PyBuffer_Release(&am);
    }
    PyBuffer_Release(&mz);
    PyBuffer_Release(&it);
    return ret;
slow_aux:
    PyBuffer_Release(&ap);
    PyBuffer_Release(&am);
slow:
    PyBuffer_Release(&mz);
    PyBuffer_Release(&it);
    Py_RETURN_NONE;
}

static PyMethodDef fast_methods[] = {
    {"setup", (PyCFunction)fast_setup, METH_O, "setup(numpy.ndarray)"},
    {"score_one", (PyCFunction)(void (*)(void))fast_score_one, METH_FASTCALL,
     "score_one(fn, handle, mz, inten, peptide, n_of_mod, max_charge, aux_pos, aux_mass)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_fast", "PyAscore.score() without ctypes", -1, fast_methods};

PyMODINIT_FUNC PyInit__fast(void) { return PyModule_Create(&fast_module); }
