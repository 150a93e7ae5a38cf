"""Row-sharded search across the GPUs of one node (one process per GPU).

The reference has no distributed path (this is new work).
The sample x feature matrix is partitioned by rows: rank g owns a contiguous
range of global ids and builds its own forest over them.  A query is answered by
every shard and the only exchange on the data path is an all-gather of the
per-shard top-k (id, distance) pairs -- nq * k * 8 bytes per rank, latency-bound
over xGMI -- after which every rank holds the merged (distance, id)-ordered top k.
`torch.distributed` with backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests with a stand-in local index.
"""
import numpy as np
import torch
import torch.distributed as dist


from .shards import merge_topk, merge_topk_exact, merge_topk_native   # noqa: E402,F401  (re-exported: tests and callers import them from here)


class ShardedSearch(object):
    """index: the local shard (AnnoyIndex-shaped).  Global id = offset[rank] + local id.

    Backend "nccl" (the product path): the exchange runs INSIDE libmorna_hip -- the handle owns an RCCL communicator
    (morna_comm_init), per-shard search, ncclAllGather and merge kernel are enqueued on the handle's stream, and this class
    only forwards; torch.distributed is used once, to hand rank 0's communicator id to the other ranks.  Backend "gloo"
    (CPU tests with a stand-in index; rehearsals of N ranks on fewer GPUs): per-shard answers through the host, gloo
    all-gather, host merge -- the same merges (shards.py), so both paths give one answer."""

    def __init__(self, index, rank, world, n_local, group=None, transport=None):
        """transport (backend "nccl" only): "library" (default) -- the communicator inside libmorna_hip; "torch" -- the same
        kernels with torch.distributed issuing the all-gathers between the library's exported halves of the exchange
        (morna_*_packed), on the handle's own stream; "auto" -- the library's, or torch's when the library's communicator
        cannot be made on some rank (the ranks agree).  `self.transport` says which one runs."""
        self.index, self.rank, self.world, self.group = index, rank, world, group
        nccl = dist.get_backend(group) == "nccl"
        self.device = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        transport = transport or "library"
        self.in_library = nccl and hasattr(index, "comm_init") and transport in ("library", "auto")
        self.transport = "gloo-host" if not nccl else "torch"
        if self.in_library:
            err = None
            try:
                try:                               # a communicator from an earlier ShardedSearch over this handle
                    have = index.comm_info(offsets=False)[:2]
                except RuntimeError:
                    have = None
                if have != (rank, world):
                    if have is not None:
                        index.comm_destroy()
                    uid = [None]
                    if rank == 0:
                        try:
                            uid = [index.comm_unique_id()]
                        except Exception as e:     # noqa: BLE001 -- rank 0 still takes part in the broadcast below
                            err = e
                    dist.broadcast_object_list(uid, src=0, group=group, device=self.device)
                    if uid[0] is None:
                        raise err or RuntimeError("rank 0 could not make a communicator id")
                    index.comm_init(uid[0], rank, world)
            except Exception as e:                 # noqa: BLE001 -- under "auto" the ranks vote below
                if transport != "auto":
                    raise
                err = e
            if transport == "auto":
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                if int(ok.item()) == 0:
                    if err is None:
                        index.comm_destroy()
                    self.in_library = False
                    self.transport_note = "library communicator failed on a rank: %s" % (err,)
        if self.in_library:
            self.transport = "library"
            self.sizes = np.diff(index.comm_info()[2]).tolist()
        else:
            sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=self.device), group=group)
            self.sizes = [int(s.item()) for s in sizes]
        self.transport_note = getattr(self, "transport_note", None)
        if self.sizes[rank] != n_local:
            raise ValueError("shard %d holds %d items, the caller says %d" % (rank, self.sizes[rank], n_local))
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.n_total = int(self.offsets[-1])

    def close(self):
        """Give the library's communicator back (before the process group goes)."""
        if self.in_library:
            self.index.comm_destroy()
            self.in_library = False

    def _all_gather_np(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in outs])

    def _global(self, ids):
        return np.where(ids >= 0, ids.astype(np.int64) + self.offsets[self.rank], -1)

    def _gather_merge(self, ids, d, k):
        """(host path) all-gather of the per-shard top-k as one [nq, 2k] int32 message per rank, then the merge"""
        gids = self._global(ids)
        if self.n_total < 2 ** 31:
            packed = np.concatenate([gids.astype(np.int32), np.ascontiguousarray(d, np.float32).view(np.int32)], axis=1)
            allp = self._all_gather_np(packed)
            all_ids = allp[:, :, :ids.shape[1]].astype(np.int64)
            all_d = np.ascontiguousarray(allp[:, :, ids.shape[1]:]).view(np.float32)
        else:
            all_ids = self._all_gather_np(gids)
            all_d = self._all_gather_np(np.ascontiguousarray(d, np.float32))
        return merge_topk_native(all_ids, all_d, k)

    def _n_each(self, n_local, n_each):
        if n_each is None:
            t = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
            dist.all_gather(t, torch.tensor([n_local], dtype=torch.int64, device=self.device), group=self.group)
            return [int(x.item()) for x in t]
        if len(n_each) != self.world or n_each[self.rank] != n_local:
            raise ValueError("n_each must list every rank's query count, this rank's being len(items)")
        return list(n_each)

    # ---- transport "torch": the library's halves of the exchange with torch.distributed's all-gather between them, all
    # of it enqueued on the handle's own stream (torch.cuda.ExternalStream): no host wait before the merged result
    def _torch_device_path(self):
        return (self.transport == "torch" and hasattr(self.index, "get_nns_by_vector_packed")
                and self.n_total < 2 ** 31 and self.world <= 64)

    def _lib_stream(self):
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.index.stream_ptr(), device=self.device)
        return self._ext_stream

    def _torch_search(self, q_ptr, nq, k, search_k, keep=None):
        with torch.cuda.stream(self._lib_stream()):
            packed = torch.empty((nq, 2 * k), dtype=torch.int32, device=self.device)
            self.index.get_nns_by_vector_packed(q_ptr, nq, k, search_k, int(self.offsets[self.rank]), packed.data_ptr())
            gathered = torch.empty((self.world, nq, 2 * k), dtype=torch.int32, device=self.device)
            dist.all_gather_into_tensor(gathered, packed, group=self.group)
            ids, d, cnt = self.index.merge_topk_packed(gathered.data_ptr(), self.world, nq, k, k)   # waits for the stream
        del keep
        return ids.astype(np.int64), d, cnt

    def _torch_exact(self, k, nq, **query):
        msg = self.index.exact_packed_bytes(nq, k)
        with torch.cuda.stream(self._lib_stream()):
            mine = torch.empty(msg, dtype=torch.uint8, device=self.device)
            self.index.exact_search_packed(mine.data_ptr(), k, int(self.offsets[self.rank]), **query)
            gathered = torch.empty((self.world, msg), dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
            ids, d, cnt = self.index.merge_exact_packed(gathered.data_ptr(), self.world, nq, k, k)
        return ids.astype(np.int64), d, cnt

    def _torch_gather_rows(self, items, n_each):
        """every rank's query rows, HBM -> xGMI -> HBM: a device tensor [sum n_each, f] (on the handle's stream)"""
        n_max, f = max(n_each), self.index.f
        with torch.cuda.stream(self._lib_stream()):
            mine = torch.zeros((n_max, f), dtype=torch.float32, device=self.device)
            if len(items):
                self.index.get_item_vectors_dev(items, mine.data_ptr())
            allq = torch.empty((self.world, n_max, f), dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(allq, mine, group=self.group)
            Q = torch.cat([allq[g, :n_each[g]] for g in range(self.world)], dim=0).contiguous()
        return Q, (mine, allq)

    def get_nns_by_vector(self, Q, k, search_k=-1):
        """Q: [nq, f] fp32, identical on every rank.  Returns merged global ids,
        distances and counts on every rank."""
        if self.in_library:
            ids, d, cnt = self.index.get_nns_by_vector_sharded(Q, k, search_k)
            return ids.astype(np.int64), d, cnt
        if self._torch_device_path() and k <= 255:
            Q = np.ascontiguousarray(Q, dtype=np.float32)
            return self._torch_search(Q.ctypes.data, Q.shape[0], k, search_k)
        ids, d, cnt = self.index.get_nns_by_vector_batch(Q, k, search_k)
        return self._gather_merge(ids, d, k)

    def exact_search(self, Q, k):
        """exact_search_nn (morna.py:681-716) over the row-sharded matrix: Q [nq, f] fp64, identical on
        every rank; per-shard exact top-k (fp64, reference order), all-gather, merge with the bisect_left
        tie rule.  Same ids and distances as one index holding all rows."""
        if self.in_library:
            ids, d, cnt = self.index.exact_search_sharded(Q, k)
            return ids.astype(np.int64), d, cnt
        if self._torch_device_path():
            Q = np.ascontiguousarray(Q, dtype=np.float64)
            return self._torch_exact(k, Q.shape[0], Q=Q)
        return self._exact_merge(*self.index.exact_search_batch(Q, k), k=k)

    def _exact_merge(self, ids, d, cnt, k):
        all_ids = self._all_gather_np(self._global(ids))
        all_d = self._all_gather_np(np.ascontiguousarray(d, np.float64))
        out_ids, out_d, counts = merge_topk_exact(all_ids, all_d, k)
        # count -1 = "the reference raises ValueError for this query" (a row whose cosine_distance has a negative
        # radicand): true of the whole matrix as soon as it is true of one shard
        failed = (self._all_gather_np(np.ascontiguousarray(cnt, np.int32)) < 0).any(axis=0)
        return out_ids, out_d, np.where(failed, -1, counts).astype(np.int32)

    def _gather_rows(self, items, n_each):
        """(host path) every rank's query rows, rank 0's first: [sum n_each, f] fp32"""
        n_max = max(n_each)
        mine = np.zeros((n_max, self.index.f), np.float32)
        if len(items):
            mine[:len(items)] = self.index.get_item_vectors(items)
        allq = self._all_gather_np(mine)               # query vectors: nq * D * 4 bytes, once
        return np.concatenate([allq[g, :n_each[g]] for g in range(self.world)], axis=0)

    def get_nns_by_local_items(self, items, k, search_k=-1, n_each=None):
        """Each rank contributes the rows of some of its own items as queries
        (the by-item form, morna.py:762); every rank gets the answers to all of them,
        ordered rank 0's queries first.  n_each: how many queries each rank contributes, when the caller
        knows (saves the exchange of the counts, a latency-bound collective per call)."""
        n_each = self._n_each(len(items), n_each)
        if self.in_library:
            ids, d, cnt = self.index.get_nns_by_item_sharded(items, k, search_k, n_each=n_each)
            return ids.astype(np.int64), d, cnt
        if self._torch_device_path() and k <= 255:
            Q, keep = self._torch_gather_rows(items, n_each)
            return self._torch_search(Q.data_ptr(), Q.shape[0], k, search_k, keep=(Q, keep))
        return self.get_nns_by_vector(self._gather_rows(items, n_each), k, search_k)

    def exact_search_by_local_items(self, items, k, n_each=None):
        """exact_search_nn with stored rows as the queries, every rank contributing rows of its own shard (configs[4]:
        every item of the index against the whole index): the fp32 row widened to fp64 is the query (morna.py:697-703)."""
        n_each = self._n_each(len(items), n_each)
        if self.in_library:
            ids, d, cnt = self.index.exact_search_by_item_sharded(items, k, n_each)
            return ids.astype(np.int64), d, cnt
        if self._torch_device_path():
            Q, keep = self._torch_gather_rows(items, n_each)
            out = self._torch_exact(k, Q.shape[0], q_dev=(Q.data_ptr(), Q.shape[0]))
            del keep
            return out
        return self.exact_search(self._gather_rows(items, n_each).astype(np.float64), k)
