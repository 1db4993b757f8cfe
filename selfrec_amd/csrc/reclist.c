/* selfrec_amd._reclist: the reference's rec_list -- {user: [(item name, score), ...]} (base/graph_recommender.py:44-58,
 * :52-53) -- built from the device ranking's (users x K) id and score arrays in one pass of C: 31.5 k lists, 630 k tuples
 * and 630 k floats at the Yelp2018 shape.  The python form of the same construction (numpy fancy index + two tolist() + a
 * zip per user) spends two thirds of its time on intermediate lists; this allocates exactly the objects that are returned.
 * CPython C API, no third-party headers; built by csrc/Makefile with the system compiler (it has no device code).
 *
 *   build(users: list[str], names: list[str], ids: buffer int32 (U x K, C order), scores: buffer float32 (U x K), k: int)
 *     -> dict in `users` order; ids must lie in [0, len(names)).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject* reclist_build(PyObject* self, PyObject* args) {
  PyObject *users, *names;
  Py_buffer ids = {0}, scores = {0};
  Py_ssize_t k;
  if (!PyArg_ParseTuple(args, "O!O!y*y*n", &PyList_Type, &users, &PyList_Type, &names, &ids, &scores, &k)) return NULL;
  PyObject* out = NULL;
  const Py_ssize_t n_users = PyList_GET_SIZE(users), n_names = PyList_GET_SIZE(names);
  if (k <= 0 || ids.len != (Py_ssize_t)(n_users * k * 4) || scores.len != ids.len) {
    PyErr_SetString(PyExc_ValueError, "reclist.build: ids / scores must be (len(users) x k) int32 / float32");
    goto done;
  }
  {
    const int32_t* id = (const int32_t*)ids.buf;
    const float* sc = (const float*)scores.buf;
#if PY_VERSION_HEX < 0x030D0000
    out = _PyDict_NewPresized(n_users);     /* (private, and internal-only from CPython 3.13 on: there the dict grows as it fills) */
#else
    out = PyDict_New();
#endif
    if (!out) goto done;
    for (Py_ssize_t u = 0; u < n_users; ++u) {
      PyObject* row = PyList_New(k);
      if (!row) { Py_CLEAR(out); goto done; }
      for (Py_ssize_t c = 0; c < k; ++c) {
        const int32_t item = id[u * k + c];
        if (item < 0 || item >= n_names) {
          Py_DECREF(row);
          Py_CLEAR(out);
          PyErr_Format(PyExc_IndexError, "reclist.build: item id %d outside the %zd names", (int)item, n_names);
          goto done;
        }
        PyObject* name = PyList_GET_ITEM(names, item);
        PyObject* score = PyFloat_FromDouble((double)sc[u * k + c]);
        PyObject* pair = score ? PyTuple_New(2) : NULL;
        if (!pair) { Py_XDECREF(score); Py_DECREF(row); Py_CLEAR(out); goto done; }
        Py_INCREF(name);
        PyTuple_SET_ITEM(pair, 0, name);
        PyTuple_SET_ITEM(pair, 1, score);
        PyList_SET_ITEM(row, c, pair);
      }
      const int rc = PyDict_SetItem(out, PyList_GET_ITEM(users, u), row);
      Py_DECREF(row);
      if (rc < 0) { Py_CLEAR(out); goto done; }
    }
  }
done:
  PyBuffer_Release(&ids);
  PyBuffer_Release(&scores);
  return out;
}

static PyMethodDef methods[] = {
    {"build", reclist_build, METH_VARARGS, "build(users, names, ids_int32, scores_float32, k) -> {user: [(name, score), ...]}"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_reclist", "rec_list construction for GraphRecommender.test()", -1, methods};
PyMODINIT_FUNC PyInit__reclist(void) { return PyModule_Create(&module); }
