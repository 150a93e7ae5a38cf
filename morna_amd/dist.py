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
import contextlib

import numpy as np
import torch
import torch.distributed as dist


from .shards import merge_topk, merge_topk_exact, merge_topk_native   # noqa: E402,F401  (re-exported: tests and callers import them from here)


class ShardedSearch(object):
    """index: the local shard (AnnoyIndex-shaped: get_nns_by_vector_batch,
    get_item_vectors).  Global id = offset[rank] + local id."""

    def __init__(self, index, rank, world, n_local, group=None):
        self.index, self.rank, self.world, self.group = index, rank, world, group
        self.device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
        sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=self.device), group=group)
        self.sizes = [int(s.item()) for s in sizes]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.n_total = int(self.offsets[-1])

    def _all_gather_np(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in outs])

    def _gather_merge(self, ids, d, k):
        """All-gather of the per-shard top-k -- the one collective on the data path -- as a single
        [nq, 2k] int32 message per rank (global ids, then the fp32 distance bits), then the merge."""
        gids = np.where(ids >= 0, ids.astype(np.int64) + self.offsets[self.rank], -1)
        if self.n_total < 2 ** 31:
            packed = np.concatenate([gids.astype(np.int32), np.ascontiguousarray(d, np.float32).view(np.int32)], axis=1)
            allp = self._all_gather_np(packed)
            all_ids = allp[:, :, :ids.shape[1]].astype(np.int64)
            all_d = np.ascontiguousarray(allp[:, :, ids.shape[1]:]).view(np.float32)
        else:
            all_ids = self._all_gather_np(gids)
            all_d = self._all_gather_np(np.ascontiguousarray(d, np.float32))
        return merge_topk_native(all_ids, all_d, k)

    def _device_path(self):
        """RCCL group and a shard that can keep its answers in HBM (libmorna_hip's packed entry points)."""
        return (self.device.type == "cuda" and hasattr(self.index, "get_nns_by_vector_packed")
                and self.n_total < 2 ** 31 and self.world <= 64)

    def _lib_stream(self):
        """The shard's own HIP stream as a torch stream: the collectives are enqueued between the library's kernels in
        stream order, so the host never waits between query rows, search, all-gather and merge."""
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.index.stream_ptr(), device=self.device)
        return self._ext_stream

    def _search_gather_merge_dev(self, q_ptr, nq, k, search_k, keep=None):
        """The data path of SURVEY.md 8(e) without a host hop: per-shard top-k written to HBM as one [nq, 2k] int32
        message (global ids, distance bits), RCCL all-gather of Q*k*8 bytes per rank, merge kernel; only the merged
        result crosses PCIe.  Everything is ordered on the library's stream; the one host wait is for the merged result.
        keep: tensors the enqueued work reads (kept alive until it has run)."""
        with torch.cuda.stream(self._lib_stream()):
            packed = torch.empty((nq, 2 * k), dtype=torch.int32, device=self.device)
            self.index.get_nns_by_vector_packed(q_ptr, nq, k, search_k, int(self.offsets[self.rank]), packed.data_ptr())
            gathered = torch.empty((self.world, nq, 2 * k), dtype=torch.int32, device=self.device)
            dist.all_gather_into_tensor(gathered, packed, group=self.group)
            ids, d, cnt = self.index.merge_topk_packed(gathered.data_ptr(), self.world, nq, k, k)   # waits for the stream
        del keep
        return ids.astype(np.int64), d, cnt

    def get_nns_by_vector(self, Q, k, search_k=-1):
        """Q: [nq, f] fp32, identical on every rank.  Returns merged global ids,
        distances and counts on every rank."""
        if self._device_path() and k <= 255:
            Q = np.ascontiguousarray(Q, dtype=np.float32)
            return self._search_gather_merge_dev(Q.ctypes.data, Q.shape[0], k, search_k)
        ids, d, cnt = self.index.get_nns_by_vector_batch(Q, k, search_k)
        return self._gather_merge(ids, d, k)

    def exact_search(self, Q, k):
        """exact_search_nn (morna.py:681-716) over the row-sharded matrix: Q [nq, f] fp64, identical on
        every rank; per-shard exact top-k (fp64, reference order), all-gather, merge with the bisect_left
        tie rule.  Same ids and distances as one index holding all rows."""
        ids, d, cnt = self.index.exact_search_batch(Q, k)
        gids = np.where(ids >= 0, ids.astype(np.int64) + self.offsets[self.rank], -1)
        all_ids = self._all_gather_np(gids)
        all_d = self._all_gather_np(np.ascontiguousarray(d, np.float64))
        out_ids, out_d, counts = merge_topk_exact(all_ids, all_d, k)
        # count -1 = "the reference raises ValueError for this query" (a row whose cosine_distance has a negative
        # radicand): true of the whole matrix as soon as it is true of one shard
        failed = (self._all_gather_np(np.ascontiguousarray(cnt, np.int32)) < 0).any(axis=0)
        return out_ids, out_d, np.where(failed, -1, counts).astype(np.int32)

    def get_nns_by_local_items(self, items, k, search_k=-1, n_each=None):
        """Each rank contributes the rows of some of its own items as queries
        (the by-item form, morna.py:762); every rank gets the answers to all of them,
        ordered rank 0's queries first.  n_each: how many queries each rank contributes, when the caller
        knows (saves the exchange of the counts, a latency-bound collective per call)."""
        if n_each is None:
            n_each = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
            dist.all_gather(n_each, torch.tensor([len(items)], dtype=torch.int64, device=self.device), group=self.group)
            n_each = [int(x.item()) for x in n_each]
        elif len(n_each) != self.world or n_each[self.rank] != len(items):
            raise ValueError("n_each must list every rank's query count, this rank's being len(items)")
        n_max = max(n_each)
        f = self.index.f
        if self.device.type == "cuda" and hasattr(self.index, "get_nns_by_vector_ptr"):
            # RCCL path: the query rows go HBM -> xGMI -> HBM, never through the host
            dev_path = self._device_path() and k <= 255 and hasattr(self.index, "get_item_vectors_dev")
            with torch.cuda.stream(self._lib_stream()) if dev_path else contextlib.nullcontext():
                even = all(n == n_max for n in n_each)
                mine = (torch.empty if even else torch.zeros)((n_max, f), dtype=torch.float32, device=self.device)
                if len(items):
                    if dev_path:
                        self.index.get_item_vectors_dev(items, mine.data_ptr())    # in stream order, no host wait
                    else:
                        self.index.get_item_vectors_into(items, mine.data_ptr())   # synchronises the library's stream
                allq = torch.empty((self.world, n_max, f), dtype=torch.float32, device=self.device)
                dist.all_gather_into_tensor(allq, mine, group=self.group)          # query vectors: nq * D * 4 bytes, once
                Q = allq.view(-1, f) if even else \
                    torch.cat([allq[g, :n_each[g]] for g in range(self.world)], dim=0).contiguous()
            if dev_path:
                return self._search_gather_merge_dev(Q.data_ptr(), Q.shape[0], k, search_k, keep=(mine, allq, Q))
            torch.cuda.current_stream().synchronize()                              # the library reads Q on its own stream
            ids, d, cnt = self.index.get_nns_by_vector_ptr(Q.data_ptr(), Q.shape[0], k, search_k)
            return self._gather_merge(ids, d, k)
        mine = np.zeros((n_max, f), np.float32)
        if len(items):
            mine[:len(items)] = self.index.get_item_vectors(items)
        allq = self._all_gather_np(mine)               # query vectors: nq * D * 4 bytes, once
        Q = np.concatenate([allq[g, :n_each[g]] for g in range(self.world)], axis=0)
        return self.get_nns_by_vector(Q, k, search_k)
