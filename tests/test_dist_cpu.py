"""Row-sharded search: the all-gather + merge path on 2 ranks (gloo, CPU).

The local shard is a numpy stand-in with the AnnoyIndex-shaped interface
ShardedSearch drives (exact angular top-k in fp32); what is under test is the
distributed logic: offsets, the top-k all-gather, the (distance, id) merge.
"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from morna_amd.dist import ShardedSearch, merge_topk


class NumpyShard(object):
    def __init__(self, X):
        self.X = np.asarray(X, np.float32)
        self.f = self.X.shape[1]

    def get_item_vectors(self, ids):
        return self.X[np.asarray(ids, np.int64)]

    def get_nns_by_vector_batch(self, Q, k, search_k=-1):
        Q = np.asarray(Q, np.float32)
        n2 = (self.X * self.X).sum(1)
        ids = np.full((len(Q), k), -1, np.int32)
        d = np.full((len(Q), k), np.inf, np.float32)
        cnt = np.zeros(len(Q), np.int32)
        for i, q in enumerate(Q):
            ppqq = n2 * float(q @ q)
            with np.errstate(divide="ignore", invalid="ignore"):
                dd = np.where(ppqq > 0, 2.0 - 2.0 * (self.X @ q) / np.sqrt(ppqq), 2.0).astype(np.float32)
            order = np.lexsort((np.arange(len(dd)), dd))[:k]
            m = len(order)
            ids[i, :m] = order
            d[i, :m] = np.sqrt(np.maximum(dd[order], 0))
            cnt[i] = m
        return ids, d, cnt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(8675309)
        X = rng.standard_normal((901, 24)).astype(np.float32)
        X[700] = X[3]                                   # a cross-shard exact tie
        bounds = [0, 400, 901]                          # ragged shards
        shard = NumpyShard(X[bounds[rank]:bounds[rank + 1]])
        ss = ShardedSearch(shard, rank, world, bounds[rank + 1] - bounds[rank])
        assert ss.offsets.tolist() == bounds and ss.n_total == 901
        Q = rng.standard_normal((7, 24)).astype(np.float32)
        Q[0] = X[3]
        ids, d, cnt = ss.get_nns_by_vector(Q, 10)
        full = NumpyShard(X).get_nns_by_vector_batch(Q, 10)
        assert ids.tolist() == full[0].astype(np.int64).tolist()
        assert np.array_equal(d, full[1])
        assert ids[0, :2].tolist() == [3, 700]          # equal distance: lower global id first
        # by-item form: rank 0 asks about 3 of its rows, rank 1 about 2 of its rows
        mine = np.array([1, 5, 9], np.int32) if rank == 0 else np.array([0, 17], np.int32)
        ids2, d2, _ = ss.get_nns_by_local_items(mine, 5)
        gl = [1, 5, 9, 400, 417]
        full2 = NumpyShard(X).get_nns_by_vector_batch(X[gl], 5)
        assert ids2.tolist() == full2[0].astype(np.int64).tolist()
        # k larger than a shard: padded slots must not leak into the merge
        tiny = ShardedSearch(NumpyShard(X[:3] if rank == 0 else X[3:5]), rank, world, 3 if rank == 0 else 2)
        ids3, _, cnt3 = tiny.get_nns_by_vector(Q[:2], 8)
        assert cnt3.tolist() == [5, 5] and (ids3[:, 5:] == -1).all()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_sharded_search_two_ranks_gloo():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


class OracleExactShard(object):
    """Shard whose exact search is the CPU oracle's restatement of exact_search_nn."""

    def __init__(self, X):
        self.X = np.asarray(X, np.float32)
        self.f = self.X.shape[1]

    def exact_search_batch(self, Q, k):
        from oracle import capi
        ids = np.full((len(Q), k), -1, np.int32)
        d = np.full((len(Q), k), np.inf, np.float64)
        cnt = np.zeros(len(Q), np.int32)
        for i, q in enumerate(Q):
            rid, rd = capi.exact_search(self.X, q, k)
            ids[i, :len(rid)] = rid
            d[i, :len(rid)] = rd
            cnt[i] = len(rid)
        return ids, d, cnt


def _exact_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)
        X = (rng.standard_normal((700, 16)) * (rng.random((700, 16)) < 0.5)).astype(np.float32)
        X[650] = X[5]                     # exact ties across the shard boundary: higher id must come first
        X[651] = X[5]
        X[20] = X[5]
        bounds = [0, 300, 700]
        ss = ShardedSearch(OracleExactShard(X[bounds[rank]:bounds[rank + 1]]), rank, world, bounds[rank + 1] - bounds[rank])
        Q = rng.standard_normal((5, 16))
        Q[0] = X[5]
        ids, d, cnt = ss.exact_search(Q, 12)
        full = OracleExactShard(X).exact_search_batch(Q, 12)
        assert ids.tolist() == full[0].astype(np.int64).tolist()
        assert d.tobytes() == full[1].tobytes()
        assert ids[0, :4].tolist() == [651, 650, 20, 5]
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_sharded_exact_search_two_ranks_gloo():
    """Config 5's shape: exact search over a row-sharded matrix equals the single-index reference scan,
    including the bisect_left tie order across shards."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exact_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_native_merge_equals_numpy_merge():
    """morna_merge_topk (C ABI, host) == the numpy (distance, id) lexsort on sorted per-shard lists with
    ties across shards, short lists, empty shards and k larger than what exists."""
    from morna_amd.dist import merge_topk_native
    rng = np.random.default_rng(5)
    for world, nq, kk, k in ((8, 300, 20, 20), (3, 50, 7, 12), (2, 9, 4, 3), (5, 40, 6, 6)):
        d = np.round(rng.random((world, nq, kk)) * 4) / 4                 # many exact ties across shards
        ids = np.empty((world, nq, kk), np.int64)
        for w in range(world):
            ids[w] = w * 1000 + rng.permuted(np.tile(np.arange(kk), (nq, 1)), axis=1)
        n_valid = rng.integers(0, kk + 1, (world, nq))
        n_valid[0, 0] = 0
        for w in range(world):                                            # each list sorted by (distance, id), empties last
            for q in range(nq):
                order = np.lexsort((ids[w, q], d[w, q]))
                ids[w, q], d[w, q] = ids[w, q][order], d[w, q][order]
                ids[w, q, n_valid[w, q]:] = -1
                d[w, q, n_valid[w, q]:] = np.inf
        d = d.astype(np.float32)
        a = merge_topk(ids, d, k)
        b = merge_topk_native(ids, d, k)
        assert a[0].tolist() == b[0].tolist()
        assert np.array_equal(a[1].astype(np.float32), b[1])
        assert a[2].tolist() == b[2].tolist()


def test_merge_topk_ties_and_padding():
    ids = np.array([[[5, 9, -1]], [[2, 7, 11]]], np.int64)          # [world=2, nq=1, k=3]
    d = np.array([[[0.1, 0.5, np.inf]], [[0.1, 0.2, 0.9]]], np.float32)
    oi, od, cnt = merge_topk(ids, d, 4)
    assert oi.tolist() == [[2, 5, 7, 9]] and cnt.tolist() == [4]
    oi, od, cnt = merge_topk(ids, d, 6)
    assert oi.tolist() == [[2, 5, 7, 9, 11, -1]] and cnt.tolist() == [5]
