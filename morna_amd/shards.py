"""Row shards of one index in ONE process: merges of per-shard answers and the AnnoyIndex-shaped front of a sharded
file set (`morna index --shards G`, index.py).  No torch here: the multi-process form (one process per GPU, RCCL) is dist.py.

The reference has no sharded path (new work).  Rank g owns the global ids
[g * ceil(N / G), (g + 1) * ceil(N / G)) and its own forest; a query is answered by every shard and the answers are
merged by (distance, id) -- annoy's own order (approximate search) or exact_search_nn's bisect_left order (exact).
"""
import numpy as np


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


def merge_topk_native(ids, dists, k):
    """merge_topk through the library (morna_merge_topk: a k-way merge of the sorted per-shard lists, C++
    on the host): the numpy lexsort above takes as long as a whole build + query step at 8 shards."""
    import ctypes as C
    from ._lib import check, lib
    world, nq, kk = ids.shape
    ids = np.ascontiguousarray(ids, np.int64)
    dists = np.ascontiguousarray(dists, np.float32)
    out_ids = np.empty((nq, k), np.int64)
    out_d = np.empty((nq, k), np.float32)
    cnt = np.empty(nq, np.int32)
    if nq:
        check(lib().morna_merge_topk(ids.ctypes.data_as(C.c_void_p), dists.ctypes.data_as(C.c_void_p), world, nq, kk, k,
                                     out_ids.ctypes.data_as(C.c_void_p), out_d.ctypes.data_as(C.c_void_p),
                                     cnt.ctypes.data_as(C.c_void_p)))
    return out_ids, out_d, cnt


def merge_topk_exact(ids, dists, k):
    """Merge of per-shard exact_search_nn results ([world, nq, k], fp64 distances): what the
    reference's bisect_left scan over ALL rows would keep -- ascending distance, and among equal
    distances the HIGHER global id first (morna.py:705-712).  NaN distances sort last."""
    world, nq, kk = ids.shape
    flat_ids = np.transpose(ids, (1, 0, 2)).reshape(nq, world * kk).astype(np.int64)
    flat_d = np.transpose(dists, (1, 0, 2)).reshape(nq, world * kk).astype(np.float64)
    empty = flat_ids < 0
    key_d = np.where(empty | np.isnan(flat_d), np.inf, flat_d)
    rank_last = (empty * 2 + (np.isnan(flat_d) & ~empty) * 1).astype(np.int64)       # real < NaN < empty
    order = np.lexsort((-flat_ids, key_d, rank_last), axis=1)[:, :k] if nq else np.zeros((0, k), np.int64)
    out_ids = np.take_along_axis(flat_ids, order, axis=1) if nq else np.zeros((0, k), np.int64)
    out_d = np.take_along_axis(flat_d, order, axis=1) if nq else np.zeros((0, k))
    if out_ids.shape[1] < k:
        pad = k - out_ids.shape[1]
        out_ids = np.concatenate([out_ids, np.full((nq, pad), -1, np.int64)], axis=1)
        out_d = np.concatenate([out_d, np.full((nq, pad), np.inf)], axis=1)
    counts = (out_ids >= 0).sum(axis=1).astype(np.int32)
    return out_ids, np.where(out_ids >= 0, out_d, np.inf), counts


class LocalShards(object):
    """The shards of `<basename>.shards.mor` loaded into this process, behind the part of the AnnoyIndex surface
    MornaSearch uses (morna.py:651, 659, 702, 762, 769): ids are GLOBAL internal ids."""

    def __init__(self, basename, dim, device=0):
        from .annoy import AnnoyIndex
        from .index import shard_basename
        with open(basename + ".shards.mor") as fh:
            world = int(fh.readline())
            self.offsets = np.array([int(t) for t in fh.readline().split()], np.int64)
        if len(self.offsets) != world + 1 or (np.diff(self.offsets) <= 0).any() or self.offsets[0] != 0:
            raise IOError(basename + ".shards.mor is damaged")
        self.f = int(dim)
        devices = list(device) if isinstance(device, (list, tuple)) else [device]
        self.shards = []
        for g in range(world):
            a = AnnoyIndex(self.f, metric="angular", device=devices[g % len(devices)])
            a.load(shard_basename(basename, g, world) + ".annoy.mor")
            if a.get_n_items() != self.offsets[g + 1] - self.offsets[g]:
                raise IOError("shard %d of %s holds %d items, %s.shards.mor says %d"
                              % (g, basename, a.get_n_items(), basename, self.offsets[g + 1] - self.offsets[g]))
            self.shards.append(a)

    def get_n_items(self):
        return int(self.offsets[-1])

    def _owner(self, i):
        if i < 0 or i >= self.offsets[-1]:
            raise IndexError("Item index %d out of range [0, %d)" % (i, self.offsets[-1]))
        g = int(np.searchsorted(self.offsets, i, side="right")) - 1
        return g, int(i - self.offsets[g])

    def get_item_vector(self, i):
        g, local = self._owner(int(i))
        return self.shards[g].get_item_vector(local)

    def _global(self, g, ids):
        return np.where(ids >= 0, ids.astype(np.int64) + self.offsets[g], -1)

    def get_nns_by_vector_batch(self, Q, n, search_k=-1):
        per = [s.get_nns_by_vector_batch(Q, n, search_k) for s in self.shards]
        return merge_topk_native(np.stack([self._global(g, p[0]) for g, p in enumerate(per)]), np.stack([p[1] for p in per]), n)

    def get_nns_by_vector(self, vector, n, search_k=-1, include_distances=False):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        if v.shape != (self.f,):
            raise IndexError("Vector has wrong length (expected %d, got %d)" % (self.f, v.size))
        ids, d, cnt = self.get_nns_by_vector_batch(v[None, :], n, search_k)
        m = int(cnt[0])
        out = [int(x) for x in ids[0, :m]]
        return (out, [float(x) for x in d[0, :m]]) if include_distances else out

    def get_nns_by_item(self, i, n, search_k=-1, include_distances=False):
        """The stored fp32 row of global item i as the query of every shard (morna.py:762, 769)."""
        g, local = self._owner(int(i))
        row = self.shards[g].get_item_vectors(np.array([local], np.int32))
        ids, d, cnt = self.get_nns_by_vector_batch(row, n, search_k)
        m = int(cnt[0])
        out = [int(x) for x in ids[0, :m]]
        return (out, [float(x) for x in d[0, :m]]) if include_distances else out

    def exact_search_batch(self, Q, n):
        per = [s.exact_search_batch(Q, n) for s in self.shards]
        ids, d, cnt = merge_topk_exact(np.stack([self._global(g, p[0]) for g, p in enumerate(per)]),
                                       np.stack([p[1] for p in per]), n)
        failed = np.stack([p[2] for p in per]).min(axis=0) < 0     # one shard's "the reference raises" fails the query
        return ids, d, np.where(failed, -1, cnt).astype(np.int32)


class DistShards(object):
    """One shard per process (torchrun: WORLD_SIZE = the number of shards of `<basename>.shards.mor`), behind the same part
    of the AnnoyIndex surface as LocalShards.  Every method is a COLLECTIVE: all ranks call it with the same arguments (the
    command line does: `morna search` under torchrun) and all receive the same answer.  The exchange is dist.ShardedSearch's:
    inside the library over RCCL when the process group's backend is "nccl", through the host over gloo otherwise."""

    def __init__(self, basename, dim, rank, world, device=0, group=None):
        from .annoy import AnnoyIndex
        from .dist import ShardedSearch
        from .index import shard_basename
        with open(basename + ".shards.mor") as fh:
            n_shards = int(fh.readline())
            self.offsets = np.array([int(t) for t in fh.readline().split()], np.int64)
        if n_shards != world or len(self.offsets) != world + 1:
            raise IOError("%s.shards.mor lists %d shards, %d processes were started" % (basename, n_shards, world))
        self.f, self.rank, self.world = int(dim), rank, world
        self.shard = AnnoyIndex(self.f, metric="angular", device=device)
        self.shard.load(shard_basename(basename, rank, world) + ".annoy.mor")
        self.search = ShardedSearch(self.shard, rank, world, self.shard.get_n_items(), group=group)
        if self.search.offsets.tolist() != self.offsets.tolist():
            raise IOError("the shards loaded by the ranks do not add up to %s.shards.mor" % basename)

    def close(self):
        self.search.close()

    def get_n_items(self):
        return int(self.offsets[-1])

    def _owner(self, i):
        if i < 0 or i >= self.offsets[-1]:
            raise IndexError("Item index %d out of range [0, %d)" % (i, self.offsets[-1]))
        g = int(np.searchsorted(self.offsets, i, side="right")) - 1
        return g, int(i - self.offsets[g])

    @staticmethod
    def _one(ids, d, cnt, include_distances):
        m = int(cnt[0])
        out = [int(x) for x in ids[0, :m]]
        return (out, [float(x) for x in d[0, :m]]) if include_distances else out

    def get_nns_by_vector(self, vector, n, search_k=-1, include_distances=False):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        if v.shape != (self.f,):
            raise IndexError("Vector has wrong length (expected %d, got %d)" % (self.f, v.size))
        return self._one(*self.search.get_nns_by_vector(v[None, :], n, search_k), include_distances=include_distances)

    def get_nns_by_item(self, i, n, search_k=-1, include_distances=False):
        """The owner of global item i hands over its stored row; every shard answers (morna.py:762, 769)."""
        g, local = self._owner(int(i))
        mine = np.array([local] if g == self.rank else [], np.int32)
        n_each = [1 if r == g else 0 for r in range(self.world)]
        return self._one(*self.search.get_nns_by_local_items(mine, n, search_k, n_each=n_each), include_distances=include_distances)

    def get_item_vector(self, i):
        g, local = self._owner(int(i))
        mine = np.array([local] if g == self.rank else [], np.int32)
        rows = self.search._gather_rows(mine, [1 if r == g else 0 for r in range(self.world)]) if not self.search.in_library \
            else self._row_via_exact(mine, g)
        return [float(x) for x in rows[0]]

    def _row_via_exact(self, mine, g):
        # (library path: the owner reads its row and the ranks exchange it through the process group)
        import torch
        import torch.distributed as dist
        row = torch.zeros(self.f, dtype=torch.float32, device=self.search.device)
        if len(mine):
            row = torch.from_numpy(self.shard.get_item_vectors(mine)[0]).to(self.search.device)
        dist.broadcast(row, src=g, group=self.search.group)
        return row.cpu().numpy()[None, :]

    def exact_search_batch(self, Q, n):
        return self.search.exact_search(np.ascontiguousarray(Q, dtype=np.float64), n)
