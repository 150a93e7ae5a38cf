"""Row-sharded search with REAL shards (libmorna_hip indexes), -m gpu.

The 8-GPU run is the driver's; what can be checked on one GPU:
  * two ranks (two processes sharing the device, gloo between them -- RCCL refuses two ranks on one GPU), each holding
    a ragged shard with its own forest, against one index over the union of the rows: exact search identical
    (ids, fp64 distances, the bisect_left tie rule across shards); approximate search equal to the merge of the two
    shards' own answers, and not worse in recall than the single index;
  * the device-resident exchange of the RCCL path (answers packed in HBM -> all-gather -> merge kernel) against the
    host merge, with ties across shards, short lists and empty slots; the 1-rank RCCL group runs it end to end in
    test_gpu_parity.py / test_gpu_configs.py.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F, T, K = 96, 10, 12
BOUNDS = [0, 2300, 5000]


def _rows():
    rng = np.random.default_rng(77)
    C = rng.standard_normal((30, F)).astype(np.float32)
    X = (C[rng.integers(0, 30, BOUNDS[-1])] + 0.5 * rng.standard_normal((BOUNDS[-1], F))).astype(np.float32)
    X[4000] = X[10]                     # the same row on both shards: equal distances across the shard boundary
    X[4001] = X[10] * np.float32(2.0)
    return X


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.dist import ShardedSearch, merge_topk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = _rows()
        shards = []
        for g in range(world):          # every rank builds both shards (builds are deterministic): rank g SERVES shard g
            a = AnnoyIndex(F)
            a.add_items(X[BOUNDS[g]:BOUNDS[g + 1]])
            a.build(T)
            shards.append(a)
        whole = AnnoyIndex(F)
        whole.add_items(X)
        whole.build(T)
        ss = ShardedSearch(shards[rank], rank, world, BOUNDS[rank + 1] - BOUNDS[rank])
        assert ss.offsets.tolist() == BOUNDS
        rng = np.random.default_rng(3)
        Q = X[rng.choice(BOUNDS[-1], 40, replace=False)] + 0.01 * rng.standard_normal((40, F)).astype(np.float32)
        Q[0] = X[10]
        # exact: the row-sharded scan IS the single scan
        eids, ed, ecnt = ss.exact_search(Q.astype(np.float64), K)
        wids, wd, wcnt = whole.exact_search_batch(Q.astype(np.float64), K)
        assert eids.tolist() == wids.astype(np.int64).tolist() and ed.tobytes() == wd.tobytes()
        assert ecnt.tolist() == wcnt.tolist()
        # rows 10, 4000 (a copy) and 4001 (twice row 10: exact in binary) are all at distance 0 from Q[0]: among equal
        # distances bisect_left leaves the HIGHER global id first, across the shard boundary too
        assert eids[0, :3].tolist() == [4001, 4000, 10]
        # approximate: what the collective returns is the merge of the shards' own answers
        for sk in (-1, 60):
            ids, d, cnt = ss.get_nns_by_vector(Q, K, sk)
            per = [s.get_nns_by_vector_batch(Q, K, sk) for s in shards]
            gi = np.stack([np.where(p[0] >= 0, p[0].astype(np.int64) + BOUNDS[g], -1) for g, p in enumerate(per)])
            want = merge_topk(gi, np.stack([p[1] for p in per]), K)
            assert ids.tolist() == want[0].tolist() and np.array_equal(np.asarray(d, np.float32), want[1])
            assert cnt.tolist() == want[2].tolist()
        ids, d, cnt = ss.get_nns_by_vector(Q, K, -1)
        wa = whole.get_nns_by_vector_batch(Q, K, -1)[0]
        rec_s = np.mean([len(set(ids[i].tolist()) & set(eids[i].tolist())) / float(K) for i in range(len(Q))])
        rec_w = np.mean([len(set(wa[i].tolist()) & set(eids[i].tolist())) / float(K) for i in range(len(Q))])
        assert rec_s >= rec_w - 0.02, (rec_s, rec_w)   # two forests inspect at least what one does
        # by-item: every rank asks about rows of its own shard
        mine = np.array([10, 11, 12], np.int32) if rank == 0 else np.array([1500, 1501], np.int32)
        ids2, d2, cnt2 = ss.get_nns_by_local_items(mine, K, -1)
        gl = [10, 11, 12, BOUNDS[1] + 1500, BOUNDS[1] + 1501]
        ids3, d3, cnt3 = ss.get_nns_by_vector(X[gl], K, -1)
        assert ids2.tolist() == ids3.tolist() and np.array_equal(d2, d3)
        assert [int(ids2[i, 0]) for i in (1, 2, 3, 4)] == gl[1:]        # a row is its own nearest neighbour
        assert sorted(ids2[0, :2].tolist()) == [10, 4000] and d2[0, 1] < 1e-3
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_shards_one_device_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_packed_device_exchange_equals_host_merge():
    """morna_get_nns_by_vector_packed + morna_merge_topk_packed: three shards' answers packed in HBM, laid out as the
    all-gather would leave them, merged on the device -- against the host k-way merge of the same answers."""
    import torch
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.dist import merge_topk_native
    X = _rows()
    bounds = [0, 7, 2300, 5000]         # the first shard is smaller than k: empty slots in its lists
    rng = np.random.default_rng(9)
    Q = np.ascontiguousarray(X[rng.choice(5000, 70, replace=False)])
    Q[1] = X[10]
    k = 20
    shards, host = [], []
    dev = torch.device("cuda", 0)
    gathered = torch.empty((3, len(Q), 2 * k), dtype=torch.int32, device=dev)
    for g in range(3):
        a = AnnoyIndex(F)
        a.add_items(X[bounds[g]:bounds[g + 1]])
        a.build(T)
        a.get_nns_by_vector_packed(Q.ctypes.data, len(Q), k, -1, bounds[g], gathered[g].data_ptr())
        ids, d, cnt = a.get_nns_by_vector_batch(Q, k, -1)
        host.append((np.where(ids >= 0, ids.astype(np.int64) + bounds[g], -1), d))
        shards.append(a)
    torch.cuda.synchronize()
    got = gathered.cpu().numpy()
    for g in range(3):                  # the message itself: global ids, then the distance bits
        assert got[g, :, :k].tolist() == host[g][0].tolist()
        assert got[g, :, k:].view(np.float32).tobytes() == host[g][1].tobytes()
    ids, d, cnt = shards[2].merge_topk_packed(gathered.data_ptr(), 3, len(Q), k, k)
    want = merge_topk_native(np.stack([h[0] for h in host]), np.stack([h[1] for h in host]), k)
    assert ids.astype(np.int64).tolist() == want[0].tolist() and d.tobytes() == want[1].tobytes()
    assert cnt.tolist() == want[2].tolist()
    assert sorted(ids[1, :2].tolist()) == [10, 4000]
    ids5, d5, cnt5 = shards[0].merge_topk_packed(gathered.data_ptr(), 3, len(Q), k, 5)   # k smaller than the lists
    assert ids5.tolist() == ids[:, :5].tolist() and (cnt5 == 5).all()
    with pytest.raises(ValueError):
        shards[0].merge_topk_packed(gathered.data_ptr(), 65, len(Q), k, 5)


def test_stream_ordered_hand_over():
    """morna_get_stream + morna_get_item_vectors_dev + the packed search without a host wait in between: work a caller
    orders on the handle's stream (here torch copies under torch.cuda.ExternalStream) sees what the library enqueued."""
    import torch
    from morna_amd.annoy import AnnoyIndex
    X = _rows()
    a = AnnoyIndex(F)
    a.add_items(X)
    a.build(T)
    dev = torch.device("cuda", 0)
    items = np.array([5, 4999, 0, 5, 1234], np.int32)
    k = 10
    ext = torch.cuda.ExternalStream(a.stream_ptr(), device=dev)
    with torch.cuda.stream(ext):
        rows = torch.empty((len(items), F), dtype=torch.float32, device=dev)
        a.get_item_vectors_dev(items, rows.data_ptr())
        packed = torch.empty((len(items), 2 * k), dtype=torch.int32, device=dev)
        a.get_nns_by_vector_packed(rows.data_ptr(), len(items), k, -1, 100, packed.data_ptr())
        msg = packed.clone()            # a torch kernel behind the library's, same stream
    a.synchronize()
    assert rows.cpu().numpy().tobytes() == X[items].tobytes()
    ids, d, cnt = a.get_nns_by_item_batch(items, k, -1)
    got = msg.cpu().numpy()
    assert got[:, :k].tolist() == (ids + 100).tolist()
    assert got[:, k:].view(np.float32).tobytes() == d.tobytes()
    with pytest.raises(IndexError):
        a.get_item_vectors_dev(np.array([5000], np.int32), rows.data_ptr())


def test_library_communicator_one_rank_through_the_c_abi_alone():
    """the RCCL communicator is owned by the handle.  No torch.distributed here: the id comes from
    morna_comm_unique_id, the handle joins a 1-rank communicator, and every sharded entry point -- per-shard search ->
    ncclAllGather -> merge kernel on the handle's stream -- returns what the plain entry point returns."""
    from morna_amd.annoy import AnnoyIndex
    X = _rows()
    a = AnnoyIndex(F)
    a.add_items(X)
    a.build(T)
    with pytest.raises(RuntimeError):
        a.get_nns_by_vector_sharded(X[:3], K, -1)               # no communicator yet
    a.comm_init(AnnoyIndex.comm_unique_id(), 0, 1)
    with pytest.raises(RuntimeError):
        a.comm_init(AnnoyIndex.comm_unique_id(), 0, 1)          # one communicator per handle
    rank, world, off = a.comm_info()
    assert (rank, world, off.tolist()) == (0, 1, [0, len(X)])
    rng = np.random.default_rng(11)
    items = rng.choice(len(X), 70, replace=False).astype(np.int32)
    Q = np.ascontiguousarray(X[items] + 0.01 * rng.standard_normal((70, F)).astype(np.float32))
    for sk in (-1, 60):
        got, want = a.get_nns_by_vector_sharded(Q, K, sk), a.get_nns_by_vector_batch(Q, K, sk)
        assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist()
        got, want = a.get_nns_by_item_sharded(items, K, sk), a.get_nns_by_item_batch(items, K, sk)
        assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist()
    got, want = a.exact_search_sharded(Q.astype(np.float64), K), a.exact_search_batch(Q.astype(np.float64), K)
    assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist()
    got = a.exact_search_by_item_sharded(items, K, [len(items)])
    want = a.exact_search_batch(X[items].astype(np.float64), K)
    by_item = a.exact_search_by_item_batch(items, K)
    for r in (got, by_item):
        assert r[0].tolist() == want[0].tolist() and r[1].tobytes() == want[1].tobytes() and r[2].tolist() == want[2].tolist()
    assert a.get_nns_by_vector_sharded(X[:0], K, -1)[0].shape == (0, K)
    with pytest.raises(ValueError):
        a.get_nns_by_item_sharded(items, K, -1, n_each=[len(items) + 1])
    with pytest.raises(IndexError):
        a.exact_search_by_item_sharded(np.array([len(X)], np.int32), K, [1])
    # the rows change -> the offsets are exchanged again
    b = AnnoyIndex(F)
    b.comm_init(AnnoyIndex.comm_unique_id(), 0, 1)
    b.add_items(X[:100])
    b.build(2)
    assert b.comm_info()[2].tolist() == [0, 100]
    a.comm_destroy()
    a.comm_destroy()                                            # idempotent
    with pytest.raises(RuntimeError):
        a.exact_search_sharded(Q.astype(np.float64), K)


def test_exact_packed_exchange_equals_host_merge():
    """morna_exact_search_packed + morna_merge_exact_packed: three shards' exact answers packed in HBM, laid end to end as
    ncclAllGather leaves them, merged on the device -- against merge_topk_exact (the host merge the gloo path uses): the
    bisect_left tie rule across shards (equal distance: higher global id first), a shard with fewer rows than k, a query
    for which the reference raises (count -1 on one shard fails it everywhere), queries as host fp64 / device fp32 / items."""
    import torch
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.dist import merge_topk_exact
    from oracle import capi
    X = _rows()
    for s in range(30):                                  # rows parallel to row 7 (exactly, in binary, for powers of two)
        X[2400 + s] = X[7] * np.float32(1.0 + 0.37 * (s + 1))
    bounds = [0, 7, 2300, 5000]
    rng = np.random.default_rng(9)
    items = rng.choice(5000, 50, replace=False)
    Q = X[items].astype(np.float64)
    Q[1] = X[10]                                         # rows 10, 4000, 4001: one distance, three shards' worth of ids
    bad = None
    for cand in range(2400, 2430):                       # a query whose cosine_distance to some row has a negative radicand
        qq = X[cand].astype(np.float64) * 1.7
        if np.isnan([capi.cosine_distance(X[r], qq) for r in [7] + list(range(2400, 2430))]).any():
            bad = qq
            break
    assert bad is not None
    Q[2] = bad
    k, nq = 20, len(Q)
    dev = torch.device("cuda", 0)
    msg = AnnoyIndex.exact_packed_bytes(nq, k)
    assert msg == (nq * (k + 1) * 4 + 7) // 8 * 8 + nq * k * 8
    gathered = torch.zeros((3, msg), dtype=torch.uint8, device=dev)
    shards, host = [], []
    for g in range(3):
        a = AnnoyIndex(F)
        a.add_items(X[bounds[g]:bounds[g + 1]])
        a.exact_search_packed(gathered[g].data_ptr(), k, bounds[g], Q=Q)
        ids, d, cnt = a.exact_search_batch(Q, k)
        host.append((np.where(ids >= 0, ids.astype(np.int64) + bounds[g], -1), d, cnt))
        shards.append(a)
    for a in shards:
        a.synchronize()
    raw = gathered.cpu().numpy()
    for g in range(3):                                   # the message itself
        m = raw[g]
        assert m[:nq * k * 4].view(np.int32).reshape(nq, k).tolist() == host[g][0].tolist()
        assert m[nq * k * 4:nq * (k + 1) * 4].view(np.int32).tolist() == host[g][2].tolist()
        assert m[msg - nq * k * 8:].view(np.float64).tobytes() == host[g][1].tobytes()
    assert host[2][2][2] == -1 and host[0][2][2] >= 0    # the parallel rows live on the last shard
    ids, d, cnt = shards[1].merge_exact_packed(gathered.data_ptr(), 3, nq, k, k)
    wi, wd, wc = merge_topk_exact(np.stack([h[0] for h in host]), np.stack([h[1] for h in host]), k)
    failed = np.stack([h[2] for h in host]).min(axis=0) < 0
    ok = ~failed
    assert ids[ok].astype(np.int64).tolist() == wi[ok].tolist() and d[ok].tobytes() == wd[ok].tobytes()
    assert cnt.tolist() == np.where(failed, -1, wc).tolist() and cnt[2] == -1
    assert ids[1, :3].tolist() == [4001, 4000, 10]
    whole = AnnoyIndex(F)
    whole.add_items(X)
    w = whole.exact_search_batch(Q, k)
    assert ids[ok].tolist() == w[0][ok].tolist() and d[ok].tobytes() == w[1][ok].tobytes() and cnt.tolist() == w[2].tolist()
    ids5, d5, cnt5 = shards[0].merge_exact_packed(gathered.data_ptr(), 3, nq, k, 5)           # k smaller than the lists
    assert ids5[ok].tolist() == ids[ok][:, :5].tolist()
    # the same message from fp32 queries in device memory and from stored rows (queries 3.. are stored rows)
    qd = torch.from_numpy(np.ascontiguousarray(X[items[3:]])).to(dev)
    m2 = torch.zeros(AnnoyIndex.exact_packed_bytes(nq - 3, k), dtype=torch.uint8, device=dev)
    m3 = torch.zeros_like(m2)
    own = [int(i) for i in items[3:] if bounds[2] <= i < bounds[3]]
    shards[2].exact_search_packed(m2.data_ptr(), k, bounds[2], q_dev=(qd.data_ptr(), nq - 3))
    shards[2].synchronize()
    ref_ids, ref_d, ref_c = shards[2].exact_search_batch(X[items[3:]].astype(np.float64), k)
    r2 = m2.cpu().numpy()
    assert r2[:(nq - 3) * k * 4].view(np.int32).reshape(nq - 3, k).tolist() == (ref_ids + bounds[2]).tolist()
    m4 = torch.zeros(AnnoyIndex.exact_packed_bytes(len(own), k), dtype=torch.uint8, device=dev)
    shards[2].exact_search_packed(m4.data_ptr(), k, bounds[2], items=np.array(own, np.int32) - bounds[2])
    shards[2].synchronize()
    sel = [j for j, i in enumerate(items[3:]) if bounds[2] <= i < bounds[3]]
    assert m4.cpu().numpy()[:len(own) * k * 4].view(np.int32).reshape(len(own), k).tolist() == (ref_ids[sel] + bounds[2]).tolist()
    del m3


def test_torch_transport_one_rank_rccl_equals_the_library_transport():
    """dist.ShardedSearch on a 1-rank RCCL process group, both device transports: the communicator inside the library
    (morna_*_sharded) and torch.distributed's all_gather_into_tensor between the library's exported halves
    (morna_get_nns_by_vector_packed / morna_merge_topk_packed, morna_exact_search_packed / morna_merge_exact_packed,
    morna_get_item_vectors_dev) on the handle's stream -- the fallback bench.py's "auto" agrees on when the library's
    communicator cannot be made.  Same answers as the plain entry points, approximate and exact, by vector and by item."""
    import torch
    import torch.distributed as dist
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.dist import ShardedSearch
    X = _rows()
    a = AnnoyIndex(F)
    a.add_items(X)
    a.build(T)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(5)
        items = rng.choice(len(X), 90, replace=False).astype(np.int32)
        Q = np.ascontiguousarray(X[items] + 0.01 * rng.standard_normal((90, F)).astype(np.float32))
        want_v = a.get_nns_by_vector_batch(Q, K, -1)
        want_i = a.get_nns_by_item_batch(items, K, 60)
        want_e = a.exact_search_batch(Q.astype(np.float64), K)
        want_ei = a.exact_search_by_item_batch(items, K)
        for transport in ("torch", "library", "auto"):
            ss = ShardedSearch(a, 0, 1, len(X), transport=transport)
            assert ss.transport == ("torch" if transport == "torch" else "library") and ss.offsets.tolist() == [0, len(X)]
            for got, want in ((ss.get_nns_by_vector(Q, K, -1), want_v),
                              (ss.get_nns_by_local_items(items, K, 60, n_each=[len(items)]), want_i),
                              (ss.exact_search(Q.astype(np.float64), K), want_e),
                              (ss.exact_search_by_local_items(items, K, n_each=[len(items)]), want_ei)):
                assert got[0].tolist() == want[0].astype(np.int64).tolist(), transport
                assert np.asarray(got[1]).tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist(), transport
            ss.close()
    finally:
        dist.destroy_process_group()
