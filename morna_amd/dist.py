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


def merge_topk(ids, dists, k):
    """ids/dists: [world, nq, k] (id -1 / dist inf = empty slot).  Returns the k
    smallest (distance, id) pairs per query as ([nq, k] ids, [nq, k] dists, [nq] counts)."""
    world, nq, kk = ids.shape
    flat_ids = np.transpose(ids, (1, 0, 2)).reshape(nq, world * kk)
    flat_d = np.transpose(dists, (1, 0, 2)).reshape(nq, world * kk)
    flat_d = np.where(flat_ids < 0, np.inf, flat_d)
    key_ids = np.where(flat_ids < 0, np.iinfo(np.int64).max, flat_ids)
    if nq == 0:
        return np.full((0, k), -1, np.int64), np.full((0, k), np.inf, flat_d.dtype), np.zeros(0, np.int32)
    # primary distance, then id: annoy's pair sort; empty slots (inf, max id) sort last
    order = np.lexsort((key_ids, flat_d), axis=1)[:, :k]
    out_ids = np.take_along_axis(flat_ids, order, axis=1)
    out_d = np.take_along_axis(flat_d, order, axis=1)
    if out_ids.shape[1] < k:                                   # fewer slots than k in total
        pad = k - out_ids.shape[1]
        out_ids = np.concatenate([out_ids, np.full((nq, pad), -1, np.int64)], axis=1)
        out_d = np.concatenate([out_d, np.full((nq, pad), np.inf, out_d.dtype)], axis=1)
    counts = (out_ids >= 0).sum(axis=1).astype(np.int32)
    out_d = np.where(out_ids >= 0, out_d, np.inf)
    return out_ids.astype(np.int64), out_d, counts


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

    def get_nns_by_vector(self, Q, k, search_k=-1):
        """Q: [nq, f] fp32, identical on every rank.  Returns merged global ids,
        distances and counts on every rank."""
        ids, d, cnt = self.index.get_nns_by_vector_batch(Q, k, search_k)
        gids = np.where(ids >= 0, ids.astype(np.int64) + self.offsets[self.rank], -1)
        all_ids = self._all_gather_np(gids)            # the one collective on the data path
        all_d = self._all_gather_np(d.astype(np.float32))
        return merge_topk(all_ids, all_d, k)

    def get_nns_by_local_items(self, items, k, search_k=-1):
        """Each rank contributes the rows of some of its own items as queries
        (the by-item form, morna.py:762); every rank gets the answers to all of them,
        ordered rank 0's queries first."""
        n_each = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(n_each, torch.tensor([len(items)], dtype=torch.int64, device=self.device), group=self.group)
        n_each = [int(x.item()) for x in n_each]
        n_max = max(n_each)
        mine = np.zeros((n_max, self.index.f), np.float32)
        if len(items):
            mine[:len(items)] = self.index.get_item_vectors(items)
        allq = self._all_gather_np(mine)               # query vectors: nq * D * 4 bytes, once
        Q = np.concatenate([allq[g, :n_each[g]] for g in range(self.world)], axis=0)
        return self.get_nns_by_vector(Q, k, search_k)
