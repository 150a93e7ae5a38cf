"""AnnoyIndex-shaped front end of the HIP library.

Mirrors the surface of ``annoy.AnnoyIndex`` that morna uses (reference call
sites in commanderson/morna morna.py): constructor 166/543/1171, ``add_item``
406/423, ``build`` 425, ``save`` 439, ``load`` 544/1172, ``get_nns_by_vector``
651/659, ``get_nns_by_item`` 762/769/1191, ``get_item_vector`` 702,
``get_n_items`` 1174 -- same positional arguments, same return shapes
(``include_distances=True`` -> ``(ids, distances)``, else a bare list).

Everything numeric happens in libmorna_hip.so on the MI355X; this module only
marshals numpy buffers through ctypes.  Batched variants (``*_batch``) expose
what the C ABI really does: many queries per launch.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, ptr


class AnnoyIndex(object):
    def __init__(self, f, metric='angular', device=0):
        if metric != 'angular':
            raise ValueError("morna uses AnnoyIndex(dim, metric='angular') only (morna.py:166); got %r" % (metric,))
        self.f = int(f)
        self._h = C.c_void_p()
        check(lib().morna_index_create(self.f, int(device), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                lib().morna_index_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # ---- items -----------------------------------------------------------
    def add_item(self, i, vector):
        v = np.ascontiguousarray(vector, dtype=np.float64)
        if v.shape != (self.f,):
            raise IndexError("Vector has wrong length (expected %d, got %d)" % (self.f, v.size))
        check(lib().morna_add_item(self._h, int(i), ptr(v)))

    def add_items(self, rows, first_id=0):
        """Bulk add_item for ids first_id.. (rows already fp32)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.f:
            raise IndexError("rows must be [n, %d]" % self.f)
        check(lib().morna_add_items_f32(self._h, int(first_id), ptr(rows), rows.shape[0]))

    def get_n_items(self):
        return int(lib().morna_get_n_items(self._h))

    def get_item_vector(self, i):
        out = np.empty(self.f, dtype=np.float32)
        check(lib().morna_get_item_vector(self._h, int(i), ptr(out)))
        return [float(x) for x in out]

    def get_item_vectors(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.empty((len(ids), self.f), dtype=np.float32)
        check(lib().morna_get_item_vectors(self._h, ptr(ids), len(ids), ptr(out)))
        return out

    def get_item_vectors_into(self, ids, out_ptr):
        """Rows of `ids` written to out_ptr: [len(ids), f] fp32 in HOST or this DEVICE's memory."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(lib().morna_get_item_vectors(self._h, ptr(ids), len(ids), C.c_void_p(int(out_ptr))))

    def get_item_vectors_dev(self, ids, out_ptr):
        """Rows of `ids` written to out_ptr ([len(ids), f] fp32, this device's memory) in the order of the handle's
        stream: no host wait."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(lib().morna_get_item_vectors_dev(self._h, ptr(ids), len(ids), C.c_void_p(int(out_ptr))))

    def stream_ptr(self):
        """The handle's HIP stream (hipStream_t as an integer), e.g. for torch.cuda.ExternalStream."""
        out = C.c_void_p()
        check(lib().morna_get_stream(self._h, C.byref(out)))
        return int(out.value or 0)

    def get_nns_by_vector_ptr(self, q_ptr, nq, n, search_k=-1):
        """get_nns_by_vector_batch for nq contiguous fp32 rows at a raw (host or device) address."""
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_get_nns_by_vector(self._h, C.c_void_p(int(q_ptr)), int(nq), int(n), int(search_k),
                                            ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def get_nns_by_vector_packed(self, q_ptr, nq, n, search_k, id_offset, packed_ptr):
        """Row-sharded search, device hand-over: answers to nq fp32 queries at q_ptr (host or device) written to
        packed_ptr (this device's memory) as the [nq, 2n] int32 message of the top-k all-gather.  Enqueued on the
        handle's stream (stream_ptr()), not waited for."""
        check(lib().morna_get_nns_by_vector_packed(self._h, C.c_void_p(int(q_ptr)), int(nq), int(n), int(search_k),
                                                   int(id_offset), C.c_void_p(int(packed_ptr))))

    def merge_topk_packed(self, gathered_ptr, world, nq, kk, n):
        """Merge of the all-gathered messages [world, nq, 2kk] (device memory) on the device."""
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_merge_topk_packed(self._h, C.c_void_p(int(gathered_ptr)), int(world), int(nq), int(kk), int(n),
                                            ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def get_items(self):
        out = np.empty((self.get_n_items(), self.f), dtype=np.float32)
        check(lib().morna_get_items(self._h, ptr(out)))
        return out

    def get_norms2(self):
        out = np.empty(self.get_n_items(), dtype=np.float32)
        check(lib().morna_get_norms2(self._h, ptr(out)))
        return out

    # ---- fused feature build (MornaIndex uses this instead of add_item) ----
    def stage_junctions(self, key_bytes, key_off, row_ptr, item_ids, cov, idf):
        key_bytes = np.ascontiguousarray(key_bytes, dtype=np.uint8)
        key_off = np.ascontiguousarray(key_off, dtype=np.int64)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        item_ids = np.ascontiguousarray(item_ids, dtype=np.int32)
        cov = np.ascontiguousarray(cov, dtype=np.int32)
        idf = np.ascontiguousarray(idf, dtype=np.float64)
        J = len(key_off) - 1
        if len(row_ptr) != J + 1 or len(idf) != J or len(item_ids) != len(cov):
            raise ValueError("stage_junctions: inconsistent array lengths")
        check(lib().morna_stage_junctions(self._h, ptr(key_bytes), ptr(key_off), J, ptr(row_ptr),
                                          ptr(item_ids), ptr(cov), ptr(idf)))

    def stage_item_order(self, order_key):
        """order_key[i] for internal id i: a key in which the lines' sample lists ascend (the external sample ids of an
        intropolis file).  A performance hint for build_features; the matrix does not depend on it."""
        order_key = np.ascontiguousarray(order_key, dtype=np.int64)
        check(lib().morna_stage_item_order(self._h, ptr(order_key), len(order_key)))

    def build_features(self, n_items):
        check(lib().morna_build_features(self._h, int(n_items)))

    def unstage_junctions(self):
        check(lib().morna_unstage_junctions(self._h))

    def hash_keys(self, key_bytes, key_off):
        key_bytes = np.ascontiguousarray(key_bytes, dtype=np.uint8)
        key_off = np.ascontiguousarray(key_off, dtype=np.int64)
        J = len(key_off) - 1
        h = np.empty(J, np.int32)
        col = np.empty(J, np.int32)
        sign = np.empty(J, np.int32)
        check(lib().morna_hash_keys(self._h, ptr(key_bytes), ptr(key_off), J, ptr(h), ptr(col), ptr(sign)))
        return h, col, sign

    # ---- forest ----------------------------------------------------------
    def build(self, n_trees, seed=0):
        check(lib().morna_build(self._h, int(n_trees), int(seed) & 0xFFFFFFFF))
        return True

    def get_n_trees(self):
        return int(lib().morna_get_n_trees(self._h))

    def forest_stats(self):
        st = _lib.ForestStats()
        check(lib().morna_get_forest_stats(self._h, C.byref(st)))
        return st.as_dict()

    def get_forest(self):
        """dict(node_rec [n_nodes,6], perm [T,N], hyperplanes [n_split,f], hp_node [n_split])."""
        st = self.forest_stats()
        rec = np.empty((st["n_nodes"], 6), np.int32)
        perm = np.empty((st["n_trees"], st["n_items"]), np.int32)
        hp = np.empty((max(st["n_split"], 1), self.f), np.float32)
        hp_node = np.empty(max(st["n_split"], 1), np.int32)
        check(lib().morna_get_forest(self._h, ptr(rec), ptr(perm), ptr(hp), ptr(hp_node)))
        return dict(node_rec=rec, perm=perm, hyperplanes=hp[:st["n_split"]], hp_node=hp_node[:st["n_split"]])

    # ---- search ------------------------------------------------------------
    def get_nns_by_vector_batch(self, Q, n, search_k=-1):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.f:
            raise IndexError("queries must be [nq, %d]" % self.f)
        nq = Q.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_get_nns_by_vector(self._h, ptr(Q), nq, int(n), int(search_k), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def get_nns_by_item_batch(self, items, n, search_k=-1):
        items = np.ascontiguousarray(items, dtype=np.int32)
        nq = items.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_get_nns_by_item(self._h, ptr(items), nq, int(n), int(search_k), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def get_nns_by_vector(self, vector, n, search_k=-1, include_distances=False):
        v = np.ascontiguousarray(vector, dtype=np.float32)   # annoy stores/queries fp32
        if v.shape != (self.f,):
            raise IndexError("Vector has wrong length (expected %d, got %d)" % (self.f, v.size))
        ids, d, cnt = self.get_nns_by_vector_batch(v[None, :], n, search_k)
        m = int(cnt[0])
        out = [int(x) for x in ids[0, :m]]
        if include_distances:
            return out, [float(x) for x in d[0, :m]]
        return out

    def get_nns_by_item(self, i, n, search_k=-1, include_distances=False):
        ids, d, cnt = self.get_nns_by_item_batch(np.array([i], np.int32), n, search_k)
        m = int(cnt[0])
        out = [int(x) for x in ids[0, :m]]
        if include_distances:
            return out, [float(x) for x in d[0, :m]]
        return out

    def exact_search_batch(self, Q, n):
        """exact_search_nn (morna.py:681-716) for fp64 queries [nq, f]."""
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        if Q.ndim != 2 or Q.shape[1] != self.f:
            raise IndexError("queries must be [nq, %d]" % self.f)
        nq = Q.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float64)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_exact_search(self._h, ptr(Q), nq, int(n), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def exact_search_by_item_batch(self, items, n):
        """exact_search_nn for stored rows as the queries (morna_exact_search_by_item): only the item numbers are sent."""
        items = np.ascontiguousarray(items, dtype=np.int32)
        nq = items.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float64)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_exact_search_by_item(self._h, ptr(items), nq, int(n), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    # ---- row-sharded search, communicator inside the library (comm.hip) --------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId: made by ONE rank and handed to the others by the caller."""
        out = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        check(lib().morna_comm_unique_id(ptr(out)))
        return out.tobytes()

    def comm_init(self, unique_id, rank, world):
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError("a communicator id has %d bytes" % _lib.COMM_ID_BYTES)
        check(lib().morna_comm_init(self._h, ptr(uid), int(rank), int(world)))

    def comm_destroy(self):
        check(lib().morna_comm_destroy(self._h))

    def comm_info(self, offsets=True):
        """(rank, world, offsets[world + 1] or None); asking for the offsets is a collective."""
        r, w = C.c_int32(), C.c_int32()
        check(lib().morna_comm_info(self._h, C.byref(r), C.byref(w), None))
        off = None
        if offsets:
            off = np.zeros(w.value + 1, np.int64)
            check(lib().morna_comm_info(self._h, C.byref(r), C.byref(w), ptr(off)))
        return r.value, w.value, off

    def get_nns_by_vector_sharded(self, Q, n, search_k=-1):
        """Q: [nq, f] fp32 (host array) or (device pointer, nq); the same on every rank.  Merged global ids."""
        if isinstance(Q, tuple):
            q_ptr, nq = C.c_void_p(int(Q[0])), int(Q[1])
        else:
            Q = np.ascontiguousarray(Q, dtype=np.float32)
            q_ptr, nq = ptr(Q), Q.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_get_nns_by_vector_sharded(self._h, q_ptr, nq, int(n), int(search_k), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def get_nns_by_item_sharded(self, local_items, n, search_k=-1, n_each=None):
        """Every rank's own items as queries; answers for all ranks' queries (rank 0's first).  n_each: every rank's
        query count -- the output arrays are sized from it (the C entry point can exchange the counts itself; a caller
        that does not know them all-gathers them first, as dist.ShardedSearch does)."""
        items = np.ascontiguousarray(local_items, dtype=np.int32)
        if n_each is None:
            if self.comm_info(offsets=False)[1] != 1:
                raise ValueError("n_each (every rank's query count) is required when there is more than one rank")
            n_each = [len(items)]
        n_each = np.ascontiguousarray(n_each, dtype=np.int64)
        nq = int(n_each.sum())
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float32)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_get_nns_by_item_sharded(self._h, ptr(items), len(items), ptr(n_each), int(n), int(search_k),
                                                  ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def exact_search_sharded(self, Q, n):
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        nq = Q.shape[0]
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float64)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_exact_search_sharded(self._h, ptr(Q), nq, int(n), ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    def exact_search_by_item_sharded(self, local_items, n, n_each):
        items = np.ascontiguousarray(local_items, dtype=np.int32)
        n_each = np.ascontiguousarray(n_each, dtype=np.int64)
        nq = int(n_each.sum())
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float64)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_exact_search_by_item_sharded(self._h, ptr(items), len(items), ptr(n_each), int(n),
                                                       ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    @staticmethod
    def exact_packed_bytes(nq, n):
        return int(lib().morna_exact_packed_bytes(int(nq), int(n)))

    def exact_search_packed(self, packed_ptr, n, id_offset, Q=None, q_dev=None, items=None):
        """Per-shard exact answers packed in HBM at packed_ptr (exact_packed_bytes(nq, n) bytes), enqueued."""
        qp = dp = ip = None
        if Q is not None:
            Q = np.ascontiguousarray(Q, dtype=np.float64)
            qp, nq = ptr(Q), Q.shape[0]
        elif q_dev is not None:
            dp, nq = C.c_void_p(int(q_dev[0])), int(q_dev[1])
        else:
            items = np.ascontiguousarray(items, dtype=np.int32)
            ip, nq = ptr(items), len(items)
        check(lib().morna_exact_search_packed(self._h, qp, dp, ip, nq, int(n), int(id_offset), C.c_void_p(int(packed_ptr))))

    def merge_exact_packed(self, gathered_ptr, world, nq, kk, n):
        ids = np.empty((nq, n), np.int32)
        d = np.empty((nq, n), np.float64)
        cnt = np.empty(nq, np.int32)
        check(lib().morna_merge_exact_packed(self._h, C.c_void_p(int(gathered_ptr)), int(world), int(nq), int(kk), int(n),
                                             ptr(ids), ptr(d), ptr(cnt)))
        return ids, d, cnt

    # ---- persistence ---------------------------------------------------------
    def save(self, fn):
        check(lib().morna_save(self._h, str(fn).encode()))
        return True

    def load(self, fn):
        check(lib().morna_load(self._h, str(fn).encode()))
        return True

    # ---- measurement -----------------------------------------------------------
    def timer_enable(self, on=True, only=None):
        """only: names of the kernel groups to bracket with events (default: all of them)."""
        mask = (1 if on else 0) if only is None else sum(2 << _lib.TIMER_NAMES.index(n) for n in only)
        check(lib().morna_timer_enable(self._h, mask))

    def timer_reset(self):
        check(lib().morna_timer_reset(self._h))

    def timers(self):
        out = {}
        for i, name in enumerate(_lib.TIMER_NAMES):
            ms, n, b = C.c_double(0), C.c_int64(0), C.c_int64(0)
            check(lib().morna_timer_read(self._h, i, C.byref(ms), C.byref(n), C.byref(b)))
            out[name] = dict(ms=ms.value, launches=n.value, bytes=b.value)
        return out

    def synchronize(self):
        check(lib().morna_synchronize(self._h))
