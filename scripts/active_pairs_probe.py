#!/usr/bin/env python3
"""How sparse is the (row tile x split node) pattern of a level of the forest build?

The split contraction (splitmm.hip) multiplies every tile of 256 rows with every split node of the level, although a row
belongs to ONE node per tree.  If the rows of a tile lie in few nodes of every tree (rows ordered so that similar rows are
neighbours), most (tile, node) blocks hold no pair that is needed.  This script builds the C3 forest, reads it back and
counts, per level and per row ordering, the node columns a tile really needs, in chunks of CH columns.

    python scripts/active_pairs_probe.py [--samples 50000] [--trees 200]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50_000)
    ap.add_argument("--junctions", type=int, default=70_000)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--features", type=int, default=3000)
    ap.add_argument("--tile", type=int, default=256)
    a = ap.parse_args()
    import torch  # noqa: F401  (its HIP runtime first)
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.index import prepare_csr
    from morna_amd.synth import SEED, synthetic_intropolis

    data = synthetic_intropolis(a.samples, J=a.junctions, seed=SEED)
    prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
    N = prep["n_items"]
    idx = AnnoyIndex(a.features)
    idx.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
    idx.stage_item_order(prep["ext_ids"])
    idx.build_features(N)
    idx.build(a.trees, seed=0)
    f = idx.get_forest()
    rec, perm = f["node_rec"], f["perm"]
    T = perm.shape[0]
    n_nodes = rec.shape[0]
    kind, tree, start, count, c0, c1 = (rec[:, i] for i in range(6))
    # levels
    is_child = np.zeros(n_nodes, bool)
    for c in (c0, c1):
        is_child[c[(kind == 0) & (c >= 0)]] = True
    level = np.full(n_nodes, -1, np.int32)
    frontier = np.nonzero(~is_child)[0]
    lv = 0
    while len(frontier):
        level[frontier] = lv
        sp = frontier[kind[frontier] == 0]
        frontier = np.concatenate([c0[sp], c1[sp]])
        frontier = frontier[frontier >= 0]
        lv += 1
    n_levels = lv
    out = {"N": int(N), "T": int(T), "tile": a.tile, "levels": []}
    # node_of[L][t][row]: the split node of level L the row is in (or -1)
    n_tiles = (N + a.tile - 1) // a.tile
    fixed = {}
    for L in range(n_levels):
        nodes = np.nonzero((level == L) & (kind == 0))[0]
        if len(nodes) == 0:
            break
        S = len(nodes)
        node_of = np.full((T, N), -1, np.int32)
        for j, nd in enumerate(nodes):
            node_of[tree[nd], perm[tree[nd], start[nd]:start[nd] + count[nd]]] = j
        # orderings available BEFORE level L's split: the level-L node of a row in a few trees (what the previous
        # level's partition has just produced)
        orders = {"natural": np.arange(N)}
        key0 = node_of[0].astype(np.int64)
        orders["tree0"] = np.argsort(key0, kind="stable")
        if T >= 3:
            key3 = (node_of[0].astype(np.int64) * 4096 + node_of[1]) * 4096 + node_of[2]
            orders["tree0,1,2"] = np.argsort(key3, kind="stable")
        if T >= 6:
            k6 = np.zeros(N, np.int64)
            for t in range(6):
                k6 = k6 * 64 + (node_of[t] + 1)
            orders["tree0..5"] = np.argsort(k6, kind="stable")
        for name in list(orders):
            if name != "natural":
                fixed["L%d:%s" % (L, name)] = orders[name]
        for name, o in fixed.items():
            if not name.startswith("L%d:" % L):
                orders[name] = o
        if T >= 12 and L >= 1:
            k12 = np.zeros(N, np.int64)
            for t in range(12):
                k12 = k12 * 32 + (node_of[t] + 1)
            fixed["L%d:tree0..11" % L] = np.argsort(k12, kind="stable")
        lvl = {"level": L, "split_nodes": int(S), "dense_blocks_256": int(n_tiles * ((S + 255) // 256)), "orders": {}}
        for name, order in orders.items():
            rank = np.empty(N, np.int64)
            rank[order] = np.arange(N)
            tile_of = rank // a.tile
            pair = (tile_of[None, :] * S + node_of).ravel()
            pair = pair[node_of.ravel() >= 0]
            act = np.unique(pair)
            per_tile = np.bincount(act // S, minlength=n_tiles)
            if name.startswith("L1:tree0..11") or name == "natural":
                # what pruning the (tile, task) pairs with few rows would leave: rows of a pair, sorted pair ids
                upair, cnt_rows = np.unique(pair, return_counts=True)
                for thr in (1, 2, 4, 8):
                    keep = upair[cnt_rows > thr]
                    pt = np.bincount(keep // S, minlength=n_tiles)
                    lvl.setdefault("prune", {})["%s thr%d" % (name, thr)] = {
                        "blocks_256": int(np.sum((pt + 255) // 256)), "pairs_to_exact_pass": int(cnt_rows[cnt_rows <= thr].sum())}
            lvl["orders"][name] = {
                "active_pairs": int(len(act)), "of_dense": float(len(act) / (n_tiles * S)),
                "blocks_256": int(np.sum((per_tile + 255) // 256)), "blocks_128": int(np.sum((per_tile + 127) // 128)),
                "max_per_tile": int(per_tile.max()), "mean_per_tile": float(per_tile.mean())}
        out["levels"].append(lvl)
        print(json.dumps(lvl), flush=True)
    return out


if __name__ == "__main__":
    main()
