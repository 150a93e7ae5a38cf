"""ctypes bindings to the C oracle (oracle/_build/libmorna_oracle.so).

TEST INFRASTRUCTURE: see oracle/__init__.py for who may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmorna_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("morna_oracle.c", "annoy_oracle.c", "Makefile")]
    if not force and os.path.exists(_SO) and all(
            os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    p = C.c_void_p
    i64, i32, u32 = C.c_int64, C.c_int32, C.c_uint32
    L.oracle_mmh3_32.restype = i32
    L.oracle_mmh3_32.argtypes = [p, i64, u32]
    L.oracle_hash_col_sign.restype = None
    L.oracle_hash_col_sign.argtypes = [p, p, i64, i32, p, p, p]
    L.oracle_index_features.restype = i64
    L.oracle_index_features.argtypes = [p, p, i64, p, p, p, i64, i64, i32, p, i64, p, p, p, p]
    L.oracle_f64_to_f32.restype = None
    L.oracle_f64_to_f32.argtypes = [p, p, i64]
    L.oracle_finalize_query.restype = None
    L.oracle_finalize_query.argtypes = [p, p, i64, p, p, i64, i32, p]
    L.oracle_cosine_distance.restype = C.c_double
    L.oracle_cosine_distance.argtypes = [p, p, i32]
    L.oracle_exact_search.restype = i64
    L.oracle_exact_search.argtypes = [p, i64, i32, i64, p, i64, p, p]
    L.annoyo_create.restype = p
    L.annoyo_create.argtypes = [C.c_int, C.c_int]
    L.annoyo_set_seed.restype = None
    L.annoyo_set_seed.argtypes = [p, u32]
    L.annoyo_destroy.restype = None
    L.annoyo_destroy.argtypes = [p]
    L.annoyo_set_items.restype = None
    L.annoyo_set_items.argtypes = [p, p, C.c_int]
    L.annoyo_build.restype = None
    L.annoyo_build.argtypes = [p, C.c_int]
    for name in ("annoyo_get_nns_by_vector",):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [p, p, C.c_int, C.c_int, p, p, p]
    L.annoyo_get_nns_by_item.restype = C.c_int
    L.annoyo_get_nns_by_item.argtypes = [p, C.c_int, C.c_int, C.c_int, p, p, p]
    for name in ("annoyo_n_nodes", "annoyo_n_roots"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [p]
    L.annoyo_root.restype = C.c_int
    L.annoyo_root.argtypes = [p, C.c_int]
    L.annoyo_split_rows.restype = i64
    L.annoyo_split_rows.argtypes = [p]
    L.annoyo_split_nodes.restype = i64
    L.annoyo_split_nodes.argtypes = [p]
    L.annoyo_norm2.restype = C.c_float
    L.annoyo_norm2.argtypes = [p, C.c_int]
    L.annoyo_node_info.restype = None
    L.annoyo_node_info.argtypes = [p, C.c_int, p]
    L.annoyo_node_vector.restype = None
    L.annoyo_node_vector.argtypes = [p, C.c_int, p]
    L.annoyo_node_items.restype = None
    L.annoyo_node_items.argtypes = [p, C.c_int, p]
    L.annoyo_dot.restype = C.c_float
    L.annoyo_dot.argtypes = [C.c_int, p, p, C.c_int]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_keys(keys):
    """list of str/bytes -> (uint8 bytes array, int64 offsets[J+1])."""
    bs = [k.encode("ascii") if isinstance(k, str) else bytes(k) for k in keys]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, off


def mmh3_32(key, seed=0):
    b = key.encode("ascii") if isinstance(key, str) else bytes(key)
    arr = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, np.uint8)
    return int(lib().oracle_mmh3_32(_ptr(arr), len(b), seed))


def hash_col_sign(keybuf, key_off, dim):
    J = len(key_off) - 1
    h = np.empty(J, np.int32)
    col = np.empty(J, np.int32)
    sign = np.empty(J, np.int32)
    lib().oracle_hash_col_sign(_ptr(keybuf), _ptr(key_off), J, dim, _ptr(h), _ptr(col), _ptr(sign))
    return h, col, sign


def index_features(keybuf, key_off, row_ptr, samples, cov, sample_count, threshold, dim, max_items=None):
    """C restatement of go_index + add_junction.  Returns a dict with the fp64
    matrix M [n_items, dim], fp32 matrix X, ext_ids, idf, freq, skipped."""
    J = len(key_off) - 1
    samples = np.ascontiguousarray(samples, dtype=np.int64)
    cov = np.ascontiguousarray(cov, dtype=np.int64)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    if max_items is None:
        max_items = int(len(np.unique(samples))) if len(samples) else 0
    M = np.zeros((max(max_items, 1), dim), dtype=np.float64)
    ext = np.zeros(max(max_items, 1), dtype=np.int64)
    idf = np.empty(J, dtype=np.float64)
    freq = np.empty(J, dtype=np.int64)
    skipped = C.c_int64(0)
    n = lib().oracle_index_features(_ptr(keybuf), _ptr(key_off), J, _ptr(row_ptr), _ptr(samples),
                                    _ptr(cov), sample_count, threshold, dim, _ptr(M), max_items,
                                    _ptr(ext), _ptr(idf), _ptr(freq), C.byref(skipped))
    if n < 0:
        raise RuntimeError("max_items too small")
    M = M[:n]
    X = np.empty((n, dim), dtype=np.float32)
    if n:
        lib().oracle_f64_to_f32(_ptr(np.ascontiguousarray(M)), _ptr(X), n * dim)
    return dict(M=M, X=X, ext_ids=ext[:n].copy(), idf=idf, freq=freq, skipped=int(skipped.value), n_items=int(n))


def finalize_query(keys, cov_sum, freq, sample_count, dim):
    buf, off = pack_keys(keys)
    cov_sum = np.ascontiguousarray(cov_sum, dtype=np.int64)
    freq = np.ascontiguousarray(freq, dtype=np.int64)
    q = np.empty(dim, dtype=np.float64)
    lib().oracle_finalize_query(_ptr(buf), _ptr(off), len(keys), _ptr(cov_sum), _ptr(freq),
                                sample_count, dim, _ptr(q))
    return q


def exact_search(X, q, k):
    X = np.ascontiguousarray(X, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n, dim = X.shape
    ids = np.empty(max(k, 1), np.int64)
    d = np.empty(max(k, 1), np.float64)
    m = lib().oracle_exact_search(_ptr(X), n, dim, dim, _ptr(q), k, _ptr(ids), _ptr(d))
    return ids[:m].copy(), d[:m].copy()


def cosine_distance(row, q):
    row = np.ascontiguousarray(row, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float64)
    return float(lib().oracle_cosine_distance(_ptr(row), _ptr(q), len(row)))


def dot(mode, x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    return float(lib().annoyo_dot(mode, _ptr(x), _ptr(y), len(x)))


class AnnoyOracle(object):
    """Restated annoy forest.  mode 0 = faithful sequential, 1 = wave order."""

    def __init__(self, f, mode=0, seed=None):
        self.f = f
        self.mode = mode
        self._h = lib().annoyo_create(f, mode)
        if seed is not None:
            lib().annoyo_set_seed(self._h, seed)
        self.n_items = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().annoyo_destroy(self._h)
            self._h = None

    def set_items(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        assert X.shape[1] == self.f
        self.n_items = X.shape[0]
        lib().annoyo_set_items(self._h, _ptr(X), X.shape[0])

    def build(self, n_trees):
        lib().annoyo_build(self._h, n_trees)

    def get_nns_by_vector(self, v, n, search_k=-1, include_distances=False, return_cand=False):
        v = np.ascontiguousarray(v, dtype=np.float32)
        ids = np.empty(max(n, 1), np.int32)
        d = np.empty(max(n, 1), np.float32)
        cand = C.c_int(0)
        m = lib().annoyo_get_nns_by_vector(self._h, _ptr(v), n, search_k, _ptr(ids), _ptr(d), C.byref(cand))
        out = ids[:m].tolist()
        res = (out, d[:m].tolist()) if include_distances else out
        return (res, cand.value) if return_cand else res

    def get_nns_by_item(self, i, n, search_k=-1, include_distances=False, return_cand=False):
        ids = np.empty(max(n, 1), np.int32)
        d = np.empty(max(n, 1), np.float32)
        cand = C.c_int(0)
        m = lib().annoyo_get_nns_by_item(self._h, i, n, search_k, _ptr(ids), _ptr(d), C.byref(cand))
        out = ids[:m].tolist()
        res = (out, d[:m].tolist()) if include_distances else out
        return (res, cand.value) if return_cand else res

    # introspection -------------------------------------------------------
    def n_nodes(self):
        return lib().annoyo_n_nodes(self._h)

    def roots(self):
        return [lib().annoyo_root(self._h, t) for t in range(lib().annoyo_n_roots(self._h))]

    def split_rows(self):
        return int(lib().annoyo_split_rows(self._h))

    def split_nodes(self):
        return int(lib().annoyo_split_nodes(self._h))

    def norm2(self):
        return np.array([lib().annoyo_norm2(self._h, i) for i in range(self.n_items)], dtype=np.float32)

    def node(self, nid):
        """dict(kind, n_desc, child0, child1, tree, level, v | items)."""
        info = np.zeros(6, np.int32)
        lib().annoyo_node_info(self._h, nid, _ptr(info))
        d = dict(kind=int(info[0]), n_desc=int(info[1]), child0=int(info[2]), child1=int(info[3]),
                 tree=int(info[4]), level=int(info[5]))
        if d["kind"] == 0:
            v = np.empty(self.f, np.float32)
            lib().annoyo_node_vector(self._h, nid, _ptr(v))
            d["v"] = v
        else:
            items = np.empty(max(d["child0"], 1), np.int32)
            lib().annoyo_node_items(self._h, nid, _ptr(items))
            d["items"] = items[:d["child0"]].copy()
        return d
