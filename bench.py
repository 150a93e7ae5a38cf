#!/usr/bin/env python3
"""Headline benchmark: morna index-build + 1000-query k-NN on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic intropolis
already resident in HBM: feature-hashed TF-IDF matrix build (hash, accumulate,
fp32 convert, row norms) + 200-tree forest build + 1000 by-item queries (k=20,
morna's default search_k=100).  Default workload = BASELINE.json configs[2]:
50k samples x 3000 features.  With --gpus N (default --scaling strong) the ONE
data set is cut into N row shards -- global idf and first-seen ids, one shard and
one forest per GPU -- so N = 1, 2, 4, 8 all index the same 50k
samples and answer the same 1000 queries ("at 50k x 3000; 1/2/4/8 GPU");
--samples 200000 --gpus 8 is configs[3].  --scaling weak gives every rank a
50k-sample data set of its own instead.  Queries are answered by every shard and
merged by an RCCL all-gather of the per-shard top-k inside the library (no other
collective on the path).

Prints ONE JSON line (rank 0).  `value` = samples indexed per second over the
whole step (build + queries), all ranks.  `roofline` is for the kernel group
that takes the most time (`rooflines` has all four): algorithmic bytes
 -- or, for the split on the matrix cores, flops -- per launch over
its HIP-event time on the library's own stream.  `cpu_baseline` times the CPU
oracle (faithful single-thread restatement of morna + annoy) on a bounded sample
of the same workload and extrapolates to the full step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)


def dpad_of(d):
    """row stride of the library: D rounded up to 256 floats"""
    return (d + 255) // 256 * 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=50_000, help="samples in total (--scaling strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: ONE data set of --samples cut into row shards, one per GPU (BASELINE's 50k x 3000 at 1/2/4/8 "
                         "GPUs; --samples 200000 --gpus 8 = configs[3]); weak: every GPU indexes its own --samples")
    ap.add_argument("--features", type=int, default=3000)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--search-k", type=int, default=100)
    ap.add_argument("--junctions", type=int, default=70_000,
                    help="junction lines; 70k gives nnz ~ 2000 per sample (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[4] shard pass reported beside the headline")
    ap.add_argument("--cpu-trees", type=int, default=24)
    ap.add_argument("--cpu-queries", type=int, default=200)
    ap.add_argument("--verify", action="store_true",
                    help="adds recall@k against the exact search and a digest of the exact answers to 64 of the queries "
                         "(the same for every --gpus N under --scaling strong: one data set, one answer)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; nccl (= RCCL) is the product path, gloo only rehearses "
                         "--gpus N on a box with fewer GPUs")
    return ap.parse_args()


def _cpu_baseline_mt(args, capi, X, items, N, D, t_feat, t_dot):
    """The same port on the box's host cores (SURVEY.md 8d "cpu-ref-mt"): trees are independent, so every
    thread builds one tree with its own oracle instance (the C calls release the GIL), then answers a share
    of the queries against it.  The feature pass stays sequential (file order is part of its definition)."""
    import threading
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 16))
    if threads < 2:
        return None
    orc = [capi.AnnoyOracle(D, mode=0, seed=1000 + i) for i in range(threads)]
    for o in orc:
        o.set_items(X)
    nq = max(1, min(args.cpu_queries, len(items)) // threads)
    t_build = [0.0] * threads
    t_query = [0.0] * threads

    def work(i):
        t0 = time.perf_counter()
        orc[i].build(1)
        t_build[i] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for it in items[i * nq:(i + 1) * nq]:
            orc[i].get_nns_by_item(int(it), args.k, args.search_k)
        t_query[i] = time.perf_counter() - t0
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    # threads trees were built in max(t_build) seconds; a query costs its single-tree time plus the root dots
    # of the other trees, and threads of them run at once
    forest = max(t_build) * args.trees / threads
    t_q = max(t_query) / nq + (args.trees - 1) * t_dot
    queries = t_q * args.queries / threads
    total = t_feat + forest + queries
    return dict(value=N / total, unit="samples/s", cores=threads, kind="port",
                sample=("oracle/ on %d threads, one oracle instance and one tree per thread (%.1fs wall), %d queries per "
                        "thread; extrapolated: features %.1fs (sequential) + forest %.1fs + queries %.1fs"
                        % (threads, wall, nq, t_feat, forest, queries)))


MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (MI355X_MICROARCH.md)


def exact_all_pairs(args, device, prep, sample_count, shard_of=None, k=20, D=8192):
    """BASELINE.json configs[4]: exact brute-force k-NN of EVERY item of a 50k x 8192 index against the index
    (morna.py:681-716 per item), through morna_exact_search_by_item -- the queries are stored rows, widened to fp64 on the
    device: only the item numbers cross PCIe.  shard_of = (g, G): one GPU's diagonal block of the 8-way row cut instead
    (its 6250 rows against themselves).  Extra key, never part of `value`: one warm pass, one timed pass; the scan runs on
    the fp32 matrix cores, so its roof is the fp32 MFMA peak."""
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.index import ParsedLines
    lines = ParsedLines.from_arrays(prep, sample_count)
    if shard_of is not None:
        lines = lines.shard(*shard_of)
    a = AnnoyIndex(D, device=device)
    lines.stage(a)
    a.build_features(lines.n_items)
    a.unstage_junctions()
    n = lines.n_items
    items = np.arange(n, dtype=np.int32)
    a.exact_search_by_item_batch(items[:min(n, 4096)], k)        # warm: workspace, code objects
    a.synchronize()
    a.timer_reset()
    a.timer_enable(True, only=["exact", "exact_scan"])
    t0 = time.perf_counter()
    ids, d, cnt = a.exact_search_by_item_batch(items, k)
    wall = time.perf_counter() - t0
    a.timer_enable(False)
    tm = a.timers()
    scan = tm["exact_scan"]
    tfl = scan["bytes"] / 1e12 / (scan["ms"] / 1e3) if scan["ms"] > 0 else 0.0
    what = ("rows %d..%d of the %d-sample set (shard %d of %d of configs[4]'s 8-way cut), its %d rows as queries"
            % (lines.id_offset, lines.id_offset + n, lines.n_items_global, shard_of[0], shard_of[1], n)) if shard_of \
        else "all %d rows of the %d x %d index as queries (configs[4] at its real size, one GPU)" % (n, n, D)
    return {"workload": what + ", k=%d, by item" % k, "rows": n, "queries": n,
            "wall_ms": 1e3 * wall, "queries_per_sec": n / wall, "scan_ms": scan["ms"], "exact_group_ms": tm["exact"]["ms"],
            "scan_launches": scan["launches"], "self_is_nearest": bool((ids[:, 0] == items).all() and (cnt == k).all()),
            "roofline": {"kernel": "exact_scan_mfma_kernel", "bound": "mfma", "achieved": tfl, "peak": MFMA_F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tfl / MFMA_F32_PEAK_TFLOPS, "flops": scan["bytes"],
                         "note": "wall_ms = the whole call on the host clock (item numbers in, ids + fp64 distances out); "
                                 "exact_group_ms = scan + selection + fp64 re-rank by HIP events"}}


def query_sweep(args, index, items, k, st):
    """Latency of the approximate search by batch size (the reference answers ONE query per process, morna.py:1345-1484 ->
    651-665 / 762-774): nq = 1, 8, 64 and the full batch, by item, each timed over repeated calls on the host clock with the
    device drained.  For nq = 1 the bytes one query has to move -- its candidates' fp16 rows for the filter, the
    hyperplanes it meets (every root + one path), the fp32 rows of the survivors are not counted -- against the HBM peak."""
    out = {}
    D, dp = args.features, dpad_of(args.features)
    for nq in (1, 8, 64, len(items)):
        sub = items[:nq]
        reps = 30 if nq <= 64 else 10
        for _ in range(3):
            index.get_nns_by_item_batch(sub, k, args.search_k)
        index.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            index.get_nns_by_item_batch(sub, k, args.search_k)
        index.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        index.timer_reset()
        index.timer_enable(True, only=["query"])
        index.get_nns_by_item_batch(sub, k, args.search_k)
        index.timer_enable(False)
        tq = index.timers()["query"]
        rows = tq["bytes"] / (4.0 * D) / max(nq, 1)          # hyperplane dots + unique candidates + 1, per query
        dots = st["n_trees"] + st["max_depth"]
        cand = max(rows - dots - 1, 0.0)
        e = {"nq": nq, "ms_per_call": ms, "us_per_query": 1e3 * ms / nq, "kernels_ms": tq["ms"],
             "rows_touched_per_query": rows,
             "path": "whole-batch fp16 contraction || root margins by query groups + one-wave descents" if (nq >= 40 and nq * min(st["leaf_capacity"], rows) >= 2 * st["n_items"])
                     else "spread: a wave per (query, tree) root margin, one wave descends, a wave per candidate (canonical fp32 dot), "
                          "one workgroup ranks" if nq < 64 else "one workgroup per query, per-candidate fp16 filter"}
        if nq == 1:
            floor_b = cand * dp * 2 + dots * dp * 4
            moved_b = cand * dp * 4 + dots * dp * 4        # the spread form reads the candidates' fp32 rows (no fp16 filter stage)
            e["byte_floor"] = {"bytes": floor_b, "us_at_hbm_peak": floor_b / (HBM_PEAK_GBS * 1e3),
                               "bytes_moved": moved_b, "kernel_us": 1e3 * tq["ms"],
                               "kernel_over_floor": 1e3 * tq["ms"] / max(floor_b / (HBM_PEAK_GBS * 1e3), 1e-9),
                               "what": "%.0f candidates x %d B (fp16 rows) + %d hyperplanes x %d B" % (cand, dp * 2, dots, dp * 4),
                               "latency_chain": "below the roots the descent pops nodes ONE AFTER THE OTHER (queue maximum -> node "
                                                "record + hyperplane -> canonical dot -> push; annoy's queue hops between trees, "
                                                "so more pops than the depth: 14 + one leaf at this query by the kernel's cycle "
                                                "stamps, DESIGN.md section 5): that chain, not the bytes, is what a lone query waits for; "
                                                "`what` estimates hyperplanes as trees + depth"}
        out["nq_%d" % nq] = e
    return out


def sharded_overhead(args, index, n_items, items, k, step):
    """What the row-sharded query path costs over the plain one when there is one shard: the same step with the queries
    going through ShardedSearch on a 1-rank RCCL group (query rows all-gathered in HBM, answers packed in HBM, all-gather,
    merge kernel).  Extra key; the N > 1 runs are the driver's."""
    import socket
    import torch
    import torch.distributed as dist
    from morna_amd.dist import ShardedSearch
    if dist.is_initialized():
        return None
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    except Exception as e:      # no RCCL in this environment: the key says so
        return {"error": str(e)[:200]}
    try:
        ss = ShardedSearch(index, 0, 1, n_items)
        n = 10

        def run(fn):
            for _ in range(2):
                fn()
            index.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            index.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        plain = run(lambda: index.get_nns_by_item_batch(items, k, args.search_k))
        shard = run(lambda: ss.get_nns_by_local_items(items, k, args.search_k, n_each=[len(items)]))
        ss.close()
        return {"plain_query_ms": plain, "sharded_query_ms": shard, "overhead_ms": shard - plain,
                "path": "inside the library (communicator owned by the handle): query rows ncclAllGather HBM -> HBM, per-shard "
                        "top-k packed in HBM, ncclAllGather, merge kernel; one host wait"}
    finally:
        dist.destroy_process_group()


def cpu_baseline(args, data, prep, items):
    """Oracle (port of morna.py + annoy, 1 thread) on a bounded sample, extrapolated."""
    from oracle import capi
    N, D = prep["n_items"], args.features
    # (1) feature accumulation on a prefix of the junction lines (~1/8 of nnz)
    J = len(data["keys"])
    Js = max(1, J // 2)
    nnz_s = int(data["row_ptr"][Js])
    buf, off = capi.pack_keys(data["keys"][:Js])
    t0 = time.perf_counter()
    ref = capi.index_features(buf, off, data["row_ptr"][:Js + 1], data["samples"][:nnz_s], data["cov"][:nnz_s],
                              data["sample_count"], 100, D, max_items=N)
    t_feat = (time.perf_counter() - t0) * (len(data["samples"]) / max(nnz_s, 1))
    del ref
    # (2) forest: cpu_trees of the 200 trees on the full matrix
    X = prep["X_host"]
    o = capi.AnnoyOracle(D, mode=0)
    o.set_items(X)
    t0 = time.perf_counter()
    o.build(args.cpu_trees)
    t_tree = (time.perf_counter() - t0) / args.cpu_trees
    # (3) queries against that forest; per-query cost grows with n_trees (one
    # hyperplane dot per root), candidates = one leaf either way at search_k=100
    nq = min(args.cpu_queries, len(items))
    t0 = time.perf_counter()
    for it in items[:nq]:
        o.get_nns_by_item(int(it), args.k, args.search_k)
    t_q_small = (time.perf_counter() - t0) / nq
    # price the root dots of the trees that were not built: the same queries against a 1-tree forest give the
    # cost of one more tree per query (all measured inside the C oracle)
    o1 = capi.AnnoyOracle(D, mode=0, seed=77)
    o1.set_items(X)
    o1.build(1)
    t0 = time.perf_counter()
    for it in items[:nq]:
        o1.get_nns_by_item(int(it), args.k, args.search_k)
    t_q_one = (time.perf_counter() - t0) / nq
    del o1
    t_dot = max(t_q_small - t_q_one, 0.0) / max(args.cpu_trees - 1, 1)
    t_q = t_q_small + (args.trees - args.cpu_trees) * t_dot
    total = t_feat + t_tree * args.trees + t_q * args.queries
    cpu_baseline.mt = _cpu_baseline_mt(args, capi, X, items, N, D, t_feat, t_dot)
    return dict(value=N / total, unit="samples/s", cores=1, kind="port",
                sample=("oracle/ (C restatement of morna.py add_junction + annoy, 1 thread): features on the first "
                        "%d of %d junction lines, %d of %d trees on the full %dx%d matrix, %d of %d queries; "
                        "extrapolated: features %.1fs + forest %.1fs + queries %.1fs"
                        % (Js, J, args.cpu_trees, args.trees, N, D, nq, args.queries, t_feat, t_tree * args.trees,
                           t_q * args.queries)))


def main():
    args = parse()
    # stdout carries ONE line, the JSON: RCCL and gloo print banners from C++ ("RCCL version ...", "[Gloo] Rank 0 is
    # connected ..."), so for the whole run file descriptor 1 points at stderr and the line is written to the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.backend == "gloo":
        # rehearsal of the N > 1 path on a box with fewer GPUs than ranks: ranks share devices and
        # the top-k all-gather goes through gloo on the host (the product path is RCCL: --backend nccl)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # MORNA_BENCH_FORCE_SHARDED=1 (under torchrun, one rank): run the sharded query path on a 1-rank
    # group, to price its collectives and hand-overs on a single-GPU box
    force_sharded = bool(os.environ.get("MORNA_BENCH_FORCE_SHARDED")) and "RANK" in os.environ
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from morna_amd import build as hip_build
    if rank == 0:
        hip_build.build()
    if world > 1:
        dist.barrier()
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.index import ParsedLines, prepare_csr, shard_bounds
    from morna_amd.synth import SEED, query_items, synthetic_intropolis
    from morna_amd.dist import ShardedSearch

    N, D, T, Q, k = args.samples, args.features, args.trees, args.queries, args.k
    strong = args.scaling == "strong"
    data = synthetic_intropolis(N, J=args.junctions, seed=SEED + (0 if strong else rank))
    prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
    n_total = prep["n_items"] if strong else None          # (weak: the sum over the ranks' own data sets, below)
    index = AnnoyIndex(D, device=local_rank)
    t_stage = time.perf_counter()   # host -> HBM copy of the junction lines: NOT part of the timed region
    if strong and world > 1:
        # ONE data set, cut by rows: this rank's lines carry the GLOBAL idf and first-seen ids (morna.py:357-382) and only
        # the entries of its own samples (morna_lines_shard); the shards stacked are the single index's matrix, bit for bit
        part = ParsedLines.from_arrays(prep, data["sample_count"]).shard(rank, world)
        t_stage = time.perf_counter()
        part.stage(index)
        n_items, nnz_local, id_offset = part.n_items, part.nnz, part.id_offset
        bounds = shard_bounds(n_total, world)
    else:
        index.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
        index.stage_item_order(prep["ext_ids"])   # the sample ids, in which the lines' lists ascend (as go_index sees them)
        n_items, nnz_local, id_offset = prep["n_items"], int(len(prep["ids"])), 0
        bounds = None
    t_stage = time.perf_counter() - t_stage
    # the exchange inside the library (its own RCCL communicator); should that communicator fail to come up on this node the
    # ranks agree on torch.distributed's all-gather between the library's two halves instead -- same kernels, same stream
    sharded = ShardedSearch(index, rank, world, n_items, transport="auto" if args.backend == "nccl" else None) \
        if (world > 1 or force_sharded) else None
    n_each = None
    if strong:
        # the SAME queries whatever the number of GPUs: Q rows of the whole index; a rank hands over those it owns
        all_items = query_items(n_total, Q)
        if bounds is not None:
            owner = np.searchsorted(np.asarray(bounds), all_items, side="right") - 1
            n_each = [int((owner == g).sum()) for g in range(world)]
            items = (all_items[owner == rank] - bounds[rank]).astype(np.int32)
            answer_order = np.concatenate([np.nonzero(owner == g)[0] for g in range(world)])   # answers come rank 0's first
        else:
            items, answer_order = all_items, np.arange(len(all_items))
            n_each = [len(items)] if sharded is not None else None
    else:
        items = query_items(n_items, Q)              # the queries this rank owns (all of them when world == 1)
        answer_order = None
        if sharded is not None:
            n_each = [len(items[g::world]) for g in range(world)]   # every rank knows every rank's share of the queries
            items = items[rank::world]
        n_total = n_items * world

    def step(split=False):
        """One pass of the hot path.  split: wait for the build before the queries start, so that the two halves can be
        timed apart (the breakdown pass); the timed region runs them back to back, as an indexing job would."""
        t0 = time.perf_counter()
        index.build_features(n_items)
        index.build(T, seed=0)
        if split:
            index.synchronize()
        t1 = time.perf_counter()
        if sharded is None:
            res = index.get_nns_by_item_batch(items, k, args.search_k)
        else:
            res = sharded.get_nns_by_local_items(items, k, args.search_k, n_each=n_each)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        index.synchronize()

    GROUPS = ("features", "two_means", "split", "query")
    # Warm-up.  Its last step is bracketed with events on every kernel group to find the group that takes the most
    # time; the timed region then brackets THAT group only (an event pair costs the stream a few microseconds of idle).
    dominant = "two_means"
    for w in range(args.warmup):
        if w == args.warmup - 1:
            index.timer_reset()
            index.timer_enable(True)
        step()
    if args.warmup > 0:
        index.timer_enable(False)
        probe = index.timers()
        dominant = max(GROUPS, key=lambda n: probe[n]["ms"])
    if world > 1:   # every rank brackets the same group
        code = torch.tensor([GROUPS.index(dominant)], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.broadcast(code, src=0)
        dominant = GROUPS[int(code.item())]
    index.timer_reset()
    index.timer_enable(True, only=[dominant] if os.environ.get("MORNA_BENCH_TIMERS", "dominant") == "dominant" else list(GROUPS))
    fence()
    t_start = time.perf_counter()
    res = None
    for _ in range(args.steps):
        _b, _q, res = step()
    fence()
    elapsed = time.perf_counter() - t_start
    index.timer_enable(False)
    timed = index.timers()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = index.forest_stats()
    # After the timed region (not part of `value`): the same step with every group bracketed, for the breakdown
    n_bd = max(1, min(args.steps, 5))
    index.timer_reset()
    index.timer_enable(True)
    tb = tq = 0.0
    for _ in range(n_bd):
        b, q, _res = step(split=True)
        tb += b
        tq += q
    tb, tq = tb * args.steps / n_bd, tq * args.steps / n_bd     # scaled to the timed region's step count for the keys below
    index.timer_enable(False)
    timers = index.timers()

    verify = None
    if args.verify and strong:
        # every rank takes part (the sharded exact search is a collective): exact answers to 64 of the queries, in the
        # queries' own order whatever the number of shards, as a digest; recall of the approximate answers against them
        import hashlib
        nv = min(64, len(answer_order))
        if sharded is not None and world > 1:
            eids, ed, _ = sharded.exact_search_by_local_items(items, k, n_each=n_each)
            back = np.argsort(answer_order)                  # answer position of query j
            eids, ed, aids = eids[back][:nv], ed[back][:nv], np.asarray(res[0])[back][:nv]
        else:
            eids, ed, _ = index.exact_search_by_item_batch(items[:nv], k)
            aids = np.asarray(res[0])[:nv]
            eids = eids.astype(np.int64)
        rec = float(np.mean([len(set(aids[i].tolist()) & set(eids[i].tolist())) / float(k) for i in range(nv)]))
        verify = {"queries": nv, "exact_digest": hashlib.sha256(np.ascontiguousarray(eids, np.int64).tobytes()
                                                               + np.ascontiguousarray(ed, np.float64).tobytes()).hexdigest(),
                  "recall_at_k_vs_exact": rec}
    out = None
    if rank == 0:
        # One roofline entry per kernel group, from the HIP-event timers of the library (events recorded on the
        # stream the kernels run on) and the algorithmic bytes of SURVEY.md 8(d) that the library counts per launch.
        # The split group is priced at one pass over the rows per level (see below), not per (row, split node).
        # kernel-name prefixes of each group as the library launches them today (features.hip, forest.hip, splitmm.hip,
        # knn.hip): the HBM-side bytes of profiles/*_traffic.json are summed over these
        kernels_of = {"features": ("morna::accumulate", "morna::transpose_norms_kernel", "morna::hash_keys_kernel", "morna::col_",
                                   "morna::line_", "morna::row_norms_kernel", "morna::flags_to_serial_kernel"),
                      "two_means": ("morna::two_means",), "tm_strip": ("morna::two_means_strip_kernel",),
                      "tm_wave": ("morna::two_means_wave_kernel",),
                      "split": ("morna::split_", "morna::rows_to_half", "morna::invert_kernel", "morna::order_", "morna::sched_"),
                      "partition": ("morna::partition_kernel", "morna::post_counts_kernel"), "query": ("morna::query_",)}
        tj = None
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_traffic.json")), reverse=True):
            with open(tpath) as fh:
                cand = json.load(fh)
            cfg = cand.get("config", {})
            if (cfg.get("samples"), cfg.get("features"), cfg.get("trees")) == (N, D, T):
                tj, tj_name = cand, os.path.relpath(tpath, ROOT)   # HBM-side bytes are only quoted for the workload they were measured on
                break

        def group(name, tms, n_steps):
            tm = tms[name]
            if tm["ms"] <= 0:
                return None
            launches = max(tm["launches"], 1)
            gbs = (tm["bytes"] / 1e9) / (tm["ms"] / 1e3)
            g = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs / HBM_PEAK_GBS, "launches": tm["launches"] // max(n_steps, 1),
                 "alg_bytes_per_launch": tm["bytes"] // launches, "ms_per_launch": tm["ms"] / launches,
                 "traffic": None}
            if name == "split":
                # SURVEY.md 8(d) prices a split at 4*D bytes per (row, split node): one row read per dot, no reuse
                # across trees (93 GB per level here: 24x what HBM can move in the time the level takes).  The
                # level-synchronous forms reuse a row across the trees on chip, so the floor of a level is ONE
                # pass over the rows of its split nodes (fp32, as stored) + its hyperplanes + one side byte per
                # (row, tree); its arithmetic floor (2*D flop per row and split node on the fp16 matrix cores) is
                # ~4x lower than that, so HBM is the roof that binds.  Both are reported.
                levels = max(tm["launches"], 1)
                rows_per_level = float(n_items)        # every row is in a split node of every tree at these levels
                min_bytes = (4.0 * dpad_of(D) * rows_per_level + float(n_items) * T) * levels \
                    + 4.0 * dpad_of(D) * st["n_split"] * n_steps
                gbs = min_bytes / 1e9 / (tm["ms"] / 1e3)
                flops = 2.0 * D * (st["split_rows"] + st["n_split"]) * n_steps      # one dot per row and split node
                mm = tms.get("split_mm", {"ms": 0, "bytes": 0, "launches": 0})
                g.update(achieved=gbs, frac=gbs / HBM_PEAK_GBS, alg_bytes_per_launch=int(min_bytes / levels),
                         alg_bytes_per_launch_no_reuse=tm["bytes"] // launches,
                         mfma_floor={"algorithmic_tflops": flops / 1e12 / (tm["ms"] / 1e3), "peak": MFMA_F16_PEAK_TFLOPS})
                if mm["ms"] > 0:
                    # the contraction alone: flops of the (row tile, task chunk) products it LAUNCHED -- counted by the
                    # library, through the per-tile task lists on the device -- over ITS event time
                    g["mfma_floor"].update(contraction_ms_per_step=mm["ms"] / max(n_steps, 1),
                                           executed_flops_per_step=mm["bytes"] / max(n_steps, 1),
                                           executed_tflops=mm["bytes"] / 1e12 / (mm["ms"] / 1e3),
                                           executed_frac_of_peak=mm["bytes"] / 1e12 / (mm["ms"] / 1e3) / MFMA_F16_PEAK_TFLOPS,
                                           executed_over_needed=mm["bytes"] / max(flops, 1.0))
            if name == "query":
                # SURVEY.md 8(d) prices a query at 4*D bytes per hyperplane dot and per candidate row (27 GB per 1000
                # queries here).  The candidate filter no longer gathers candidate rows: one contraction of all
                # queries against the whole fp16 image reads the matrix ONCE, so the group's longest kernel is bound
                # by the matrix cores, and that is the roof quoted.  Beside it: the bytes the group is defined to move
                # (fp16 matrix once + scores out and back in for the candidates + the forest's hyperplanes once -- the
                # traversals' re-reads of them are served by L2 -- + fp32 rows of the survivors, ~2k per query) against
                # the HBM peak.
                f = tms.get("query_filter", {"ms": 0, "bytes": 0})
                if f["ms"] > 0:
                    tfl = f["bytes"] / 1e12 / (f["ms"] / 1e3)
                    cand_rows = tm["bytes"] / (4.0 * D)            # hyperplane dots + candidates + 1, all queries, all steps
                    defined = n_steps * (2.0 * dpad_of(D) * n_items + 4.0 * Q * n_items) + 4.0 * cand_rows \
                        + n_steps * (Q * 4.0 * dpad_of(D) * (2 * k + 1) + 4.0 * dpad_of(D) * st["n_split"])
                    g.update(bound="mfma", achieved=tfl, peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=tfl / MFMA_F16_PEAK_TFLOPS,
                             ms_per_launch=f["ms"] / max(f["launches"], 1), flops_per_launch=f["bytes"] // max(f["launches"], 1),
                             group_ms_per_launch=tm["ms"] / launches,
                             hbm={"defined_bytes_per_launch": int(defined / launches), "achieved": defined / 1e9 / (tm["ms"] / 1e3),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": defined / 1e9 / (tm["ms"] / 1e3) / HBM_PEAK_GBS,
                                  "survey_8d_bytes_per_launch": tm["bytes"] // launches})
                    del g["alg_bytes_per_launch"]
            if tj:
                ks = [v for kk, v in tj["kernels"].items() if kk.startswith(kernels_of[name])]
                if ks:   # bytes past L2 per timed launch group, from separate rocprofv3 --pmc passes
                    g["traffic"] = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in ks) / max(tj.get("steps", 1), 1) \
                        / max(tm["launches"] // max(n_steps, 1), 1)
                    g["traffic_source"] = tj_name + " (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            if g["frac"] > 1.0:
                # algorithmic bytes over time can only pass the HBM peak when the bytes never came from HBM: a matrix that
                # fits the 256-MB Infinity Cache (rehearsal sizes).  At the benchmark's own sizes it means wrong accounting.
                if 4.0 * dpad_of(D) * n_items >= 512e6 or g["bound"] != "hbm":
                    raise SystemExit("bench.py: roofline fraction %.2f > 1 for %s -- the bytes or the peak are wrong" % (g["frac"], name))
                g["served_from_cache"] = "the %d x %d matrix (%.0f MB) fits the Infinity Cache: this is not an HBM rate" % (
                    n_items, D, 4e-6 * dpad_of(D) * n_items)
            return g
        groups = {n: group(n, timers, n_bd) for n in GROUPS + ("tm_strip", "tm_wave")}   # the two two_means kernels beside their group
        groups = {n: g for n, g in groups.items() if g}
        notes = {"two_means": "one chain of 200 dependent steps per split node (annoy's two_means): bound by the latency of "
                              "that chain at shallow levels and by the random-row gather from HBM at deep ones",
                 "split": "a level's sides as one fp16 MFMA contraction that filters + exact fp32 dots for the ~0.5% it leaves open; "
                          "bytes = one pass over the fp32 rows per level + hyperplanes + side bytes (rows reused across trees on chip)",
                 "features": "fp64 accumulation in file order in LDS tiles; the nnz stream is read once per sample tile",
                 "query": "traversal || whole-batch fp16 filter contraction on the matrix cores, then fp32 dots for the survivors"}
        notes["tm_strip"] = "two_means at levels of <= 2 nodes per CU: four waves per node, a 200-step latency chain"
        notes["tm_wave"] = "two_means at the deep levels: one wave per node, bound by the gather of random 12-KB rows from HBM"
        for n, g in groups.items():
            g["note"] = notes[n]
        roofline = dict(group(dominant, timed, args.steps), note=notes[dominant], measured="HIP events in the timed region")
        total_samples = n_total * args.steps
        which = "configs[3]" if (strong and N == 200_000 and world == 8) else "configs[2]" if N == 50_000 else "a size of its own"
        if strong:
            workload = ("synthetic intropolis %d samples x %d features%s, %d trees, %d by-item queries, k=%d, search_k=%d "
                        "(BASELINE.json %s)" % (N, D, " as ONE data set cut into %d row shards" % world if world > 1 else "",
                                                T, Q, k, args.search_k, which))
        else:
            workload = ("synthetic intropolis %d samples/GPU x %d features, %d trees, %d by-item queries, k=%d, search_k=%d "
                        "(weak scaling of BASELINE.json configs[2]: a data set per GPU)" % (N, D, T, Q, k, args.search_k))
        out = {
            "metric": "samples indexed/sec (index build + %d queries, k=%d) at %dk x %d" % (Q, k, N // 1000, D),
            "value": total_samples / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "samples_total": n_total, "samples_per_gpu": n_items, "features": D, "n_trees": T, "queries": Q, "k": k,
                       "search_k": args.search_k, "nnz_per_gpu": nnz_local, "junction_lines": int(len(prep["idf"])),
                       "parallelism": "rows sharded over %d GPU(s)%s, RCCL all-gather of per-shard top-k%s"
                                      % (world, " (one data set, global idf and ids)" if strong else " (a data set per GPU)",
                                         "" if sharded is None else " -- transport: %s%s" % (
                                             sharded.transport, " (%s)" % sharded.transport_note if sharded.transport_note else ""))},
            "index_samples_per_sec": n_total * args.steps / tb,
            "queries_per_sec": Q * args.steps / tq,
            # if the junction lines had to cross PCIe on every step (pageable host buffers, measured once)
            "stage_ms": 1e3 * t_stage,
            "samples_per_sec_pcie_inclusive": n_total / (elapsed / args.steps + t_stage),
            # build / query halves: from the breakdown pass (the build drained before the queries start), not the timed region
            "build_ms_per_step": 1e3 * tb / args.steps, "query_ms_per_step": 1e3 * tq / args.steps,
            # breakdown: %d extra steps AFTER the timed region, every group bracketed with events
            "kernel_ms_per_step": {n: round(v["ms"] / n_bd, 3) for n, v in timers.items()},
            "forest": {"n_nodes": st["n_nodes"], "n_split": st["n_split"], "max_depth": st["max_depth"],
                       "split_rows": st["split_rows"], "split_attempts": st["split_attempts"],
                       "fallback_nodes": st["fallback_nodes"]},
            "roofline": roofline,
            "rooflines": groups,
        }
        if verify is not None:
            out["verify"] = verify
        if world == 1 and not args.no_extras:
            out["query_sweep"] = query_sweep(args, index, items, k, st)
            if strong and N == 50_000:
                out["exact_all_pairs"] = exact_all_pairs(args, local_rank, prep, data["sample_count"])
                out["exact_all_pairs_shard"] = exact_all_pairs(args, local_rank, prep, data["sample_count"], shard_of=(0, 8))
            if sharded is None:
                out["sharded_path_at_world_1"] = sharded_overhead(args, index, n_items, items, k, step)
        if world == 1 and not args.no_cpu_baseline:
            prep["X_host"] = index.get_items()
            out["cpu_baseline"] = cpu_baseline(args, data, prep, items)
            out["cpu_baseline_mt"] = getattr(cpu_baseline, "mt", None)   # extra: the port on all host cores
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        if sharded is not None:
            sharded.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
