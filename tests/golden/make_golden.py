#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ (run once, in the authoring
container; outputs are committed).

Sources of truth:
  * the inline fixture strings and expected neighbour orderings of the
    reference's embedded unit tests (/root/reference/morna.py:1077-1127,
    1176-1187, 1267-1278, 1312-1323) -- DATA copied into
    morna_embedded_fixtures.json by hand below;
  * /root/reference/tests/tiny_intropolis.tsv -- a data file the reference's
    tests hold, copied verbatim;
  * sklearn.utils.murmurhash3_32 (same published MurmurHash3_x86_32 as mmh3;
    reproduces mmh3's documented "foo" -> -156908512, "hello" -> 613153351);
  * oracle/morna_ref.py (py3 restatement of morna.py) for the derived matrices
    and the exact-search known answers.

The reference itself cannot be imported (Python 2 syntax; annoy/mmh3/BitVector
absent), so no vector here comes from running it.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import morna_ref  # noqa: E402

GENERIC = [
    'chr10\t100138610\t100139562\t-\tGC\tAG\t1\t1\n',
    'chr10\t100347256\t100362229\t-\tGC\tAG\t2\t3\n',
    'chr10\t100526502\t100529392\t-\tAT\tAC\t3\t1\n',
    'chr10\t100526535\t100527418\t+\tAT\tAC\t4\t1\n',
    'chr10\t100548141\t100548202\t+\tGC\tAG\t5\t1\n',
    'chr10\t100947776\t101440795\t-\tGC\tAG\t6\t1\n',
    'chr10\t100962117\t100963108\t+\tGC\tAG\t7\t1\n',
    'chr10\t100979320\t100982343\t+\tGC\tAG\t8\t1\n',
    'chr10\t101003908\t101007637\t-\tGC\tAG\t9\t1\n',
    'chr10\t101035534\t101036515\t-\tGT\tAG\t10\t1\n',
    'chr10\t101039923\t101040322\t-\tGC\tAG\t1,2,3,4\t1,1,1,1\n',
    'chr10\t101354229\t101407426\t-\tAT\tAC\t5\t1\n',
    'chr10\t101579666\t101583591\t-\tAT\tAC\t6\t1\n',
    'chr10\t101669187\t102046680\t+\tAT\tAC\t7,8\t1,1\n',
    'chr10\t101885932\t101918495\t-\tGC\tAG\t9\t1\n',
    'chr10\t102078553\t102078686\t-\tGC\tAG\t10\t1\n',
    'chr10\t102161920\t102162275\t+\tGC\tAG\t1,2,3,4,5,6,7,8,9,10\t2,1,1,2,2,1,2,3,8,2\n',
    'chr10\t102424521\t102425341\t-\tGC\tAG\t1\t1\n',
    'chr10\t102479334\t102813545\t+\tGT\tAG\t2\t1\n',
    'chr10\t102594066\t102615267\t+\tGT\tAG\t3,4,5,6,7,8,9,10\t1,4,1,1,1,7,1,1\n',
]
LOSSY = [
    'chr10\t100138610\t100139562\t-\tGC\tAG\t10\t1\n',
    'chr10\t100347256\t100362229\t-\tGC\tAG\t9\t3\n',
    'chr10\t100526502\t100529392\t-\tAT\tAC\t8\t1\n',
    'chr10\t100526535\t100527418\t+\tAT\tAC\t7\t1\n',
    'chr10\t100548141\t100548202\t+\tGC\tAG\t6\t1\n',
    'chr10\t100947776\t101440795\t-\tGC\tAG\t5\t1\n',
    'chr10\t100962117\t100963108\t+\tGC\tAG\t4\t1\n',
    'chr10\t100979320\t100982343\t+\tGC\tAG\t3\t1\n',
    'chr10\t101003908\t101007637\t-\tGC\tAG\t2\t1\n',
    'chr10\t101035534\t101036515\t-\tGT\tAG\t1\t1\n',
    'chr10\t101039923\t101040322\t-\tGC\tAG\t10,9,8,7\t1,1,1,1\n',
    'chr10\t101354229\t101407426\t-\tAT\tAC\t6\t1\n',
    'chr10\t101579666\t101583591\t-\tAT\tAC\t5\t1\n',
    'chr10\t101669187\t102046680\t+\tAT\tAC\t4,3\t1,1\n',
    'chr10\t101885932\t101918495\t-\tGC\tAG\t2\t1\n',
    'chr10\t102078553\t102078686\t-\tGC\tAG\t1\t1\n',
    'chr10\t102161920\t102162275\t+\tGC\tAG\t10,9,8,7,6,5,4\t2,1,1,2,2,1,2\n',
    'chr10\t102424521\t102425341\t-\tGC\tAG\t10\t1\n',
    'chr10\t102479334\t102813545\t+\tGT\tAG\t9\t1\n',
    'chr10\t102594066\t102615267\t+\tGT\tAG\t4,3,2,1\t1,7,1,1\n',
]
EXPECTED = {
    # morna.py:1151-1193 test_simple_indexing: sample_count=10, threshold=1, 10 items
    "simple": dict(input="generic", sample_count=10, sample_threshold=1, n_items=10, orderings=[
        [0, 2, 3, 1, 4, 5, 6, 7, 8, 9],
        [1, 2, 3, 0, 4, 5, 6, 7, 8, 9],
        [2, 3, 0, 1, 7, 6, 4, 5, 8, 9],
        [3, 7, 2, 0, 1, 6, 4, 5, 8, 9],
        [4, 7, 3, 2, 6, 5, 8, 9, 0, 1],
        [5, 7, 3, 2, 6, 4, 8, 9, 0, 1],
        [6, 7, 3, 2, 4, 5, 8, 9, 0, 1],
        [7, 6, 3, 2, 4, 5, 8, 9, 0, 1],
        [8, 7, 3, 2, 6, 4, 5, 9, 0, 1],
        [9, 7, 3, 2, 6, 4, 5, 8, 0, 1]]),
    # morna.py:1242-1284 test_lossy_indexing: threshold=4, 10 items
    "lossy": dict(input="lossy", sample_count=10, sample_threshold=4, n_items=10, orderings=[
        [0, 3, 1, 2, 4, 5, 6, 7, 8, 9],
        [1, 2, 0, 3, 4, 5, 6, 7, 8, 9],
        [1, 2, 0, 3, 4, 5, 6, 7, 8, 9],
        [0, 3, 1, 2, 4, 5, 6, 7, 8, 9],
        [4, 5, 0, 3, 6, 1, 2, 7, 8, 9],
        [4, 5, 0, 3, 6, 1, 2, 7, 8, 9],
        [6, 8, 9, 7, 4, 5, 0, 3, 1, 2],
        [7, 8, 9, 6, 0, 1, 2, 3, 4, 5],
        [8, 9, 7, 6, 0, 1, 2, 3, 4, 5],
        [8, 9, 7, 6, 0, 1, 2, 3, 4, 5]]),
    # morna.py:1286-1330 test_lose_sample_indexing: threshold=6, 7 items
    "lose_sample": dict(input="lossy", sample_count=10, sample_threshold=6, n_items=7, orderings=[
        [0, 1, 2, 3, 4, 5, 6]] * 7),
}

# morna.py:1128-1139 (metafile text) and 1221-1233 (test_metadata_indexing: search_member_n(1, 10, 100,
# include_distances=False, meta_db=True) on the "simple" index) -- fetchone() tuples as lists
META = [
    '10\tSample 10\tThe 10th sample\n',
    '1\tSample 1\tThe 1st sample\n',
    '2\tSample 2\tThe 2nd sample\n',
    '3\tSample 3\tThe 3rd sample\n',
    '4\tSample 4\tThe 4th sample\n',
    '5\tSample 5\tThe 5th sample\n',
    '6\tSample 6\tThe 6th sample\n',
    '7\tSample 7\tThe 7th sample\n',
    '8\tSample 8\tThe 8th sample\n',
    '9\tSample 9\tThe 9th sample\n',
]
META_EXPECTED = dict(query_sample_id=1, ids=[0, 2, 3, 1, 4, 5, 6, 7, 8, 9], keywords=[
    ['Sample 1\tThe 1st sample\n'], ['Sample 3\tThe 3rd sample\n'], ['Sample 4\tThe 4th sample\n'],
    ['Sample 2\tThe 2nd sample\n'], ['Sample 5\tThe 5th sample\n'], ['Sample 6\tThe 6th sample\n'],
    ['Sample 7\tThe 7th sample\n'], ['Sample 8\tThe 8th sample\n'], ['Sample 9\tThe 9th sample\n'],
    ['Sample 10\tThe 10th sample\n']])


def main():
    with open(os.path.join(HERE, "morna_embedded_fixtures.json"), "w") as fh:
        json.dump(dict(
            source="/root/reference/morna.py:1077-1127 (inputs), 1176-1187, 1267-1278, 1312-1323 (expected)",
            note=("expected orderings were captured by the reference authors at a collision-free "
                  "feature dimension (3000); rows containing exact distance ties are pinned only "
                  "up to tie order (SURVEY.md section 4)"),
            generic=GENERIC, lossy=LOSSY, expected=EXPECTED, meta=META, meta_expected=META_EXPECTED,
            sample_count_expected=10), fh, indent=1)

    ref_tiny = "/root/reference/tests/tiny_intropolis.tsv"
    if os.path.exists(ref_tiny):
        shutil.copyfile(ref_tiny, os.path.join(HERE, "tiny_intropolis.tsv"))

    # ---- murmur3 known answers -------------------------------------------
    from sklearn.utils import murmurhash3_32
    keys = ["", "a", "ab", "abc", "abcd", "abcde", "foo", "hello", "chr1 14830 14929",
            "chr1 14830 14970", "chr1 15039 15796", "chrX 1 2", "chr10 102594066 102615267"]
    rng = np.random.default_rng(8675309)
    for _ in range(200):
        c = int(rng.integers(1, 23))
        s = int(rng.integers(10_000, 240_000_000))
        e = s + int(rng.integers(60, 500_000))
        keys.append("chr%d %d %d" % (c, s, e))
    for line in GENERIC + LOSSY:
        keys.append(morna_ref.tokenize_line(line)[0])
    vec = []
    for k in keys:
        h = int(murmurhash3_32(k, seed=0, positive=False))
        assert h == morna_ref.mmh3_hash(k), k
        vec.append([k, h])
    doc = {"foo": -156908512, "hello": 613153351}   # mmh3's documented answers
    for k, h in doc.items():
        assert int(murmurhash3_32(k, seed=0, positive=False)) == h
    with open(os.path.join(HERE, "murmur3_vectors.json"), "w") as fh:
        json.dump(dict(generator="sklearn.utils.murmurhash3_32(key, seed=0, positive=False)",
                       documented=doc, vectors=vec), fh, indent=0)

    # ---- feature matrices of the inline fixtures ----------------------------
    lines = dict(generic=GENERIC, lossy=LOSSY)
    mats = {}
    for name, spec in EXPECTED.items():
        for D in (40, 128, 3000):
            idx = morna_ref.go_index_lines(lines[spec["input"]], D, spec["sample_count"],
                                           spec["sample_threshold"])
            assert idx.new_internal_id == spec["n_items"]
            mats["%s_D%d_f64" % (name, D)] = idx.matrix64()
            mats["%s_D%d_f32" % (name, D)] = idx.matrix32()
            mats["%s_D%d_ext_ids" % (name, D)] = np.array(
                sorted(idx.internal_id_map, key=idx.internal_id_map.get), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "embedded_feature_matrices.npz"), **mats)

    # ---- tiny_intropolis at D=128 (BASELINE config 1) -----------------------
    with open(os.path.join(HERE, "tiny_intropolis.tsv")) as fh:
        tlines = fh.readlines()
    idx = morna_ref.go_index_lines(tlines, 128, None, 100)
    np.savez_compressed(os.path.join(HERE, "tiny_intropolis_D128.npz"),
                        X=idx.matrix32(), M=idx.matrix64(),
                        ext_ids=np.array(sorted(idx.internal_id_map, key=idx.internal_id_map.get), dtype=np.int64),
                        sample_count=np.int64(idx.sample_count))

    # ---- exact search known answers on a seeded 512 x 128 synthetic --------
    rng = np.random.default_rng(8675309)
    X = (rng.standard_normal((512, 128)) * (rng.random((512, 128)) < 0.3)).astype(np.float32)
    X[100] = X[7]            # exact duplicates: exercises the bisect_left tie rule
    X[101] = X[7]
    X[200] = 2.0 * X[9]      # scaled duplicate
    X[300] = 0.0             # zero row: ppqq == 0 branch
    Q = rng.standard_normal((8, 128))
    Q[1] = X[7].astype(np.float64)
    Q[2] = X[9].astype(np.float64)
    s = morna_ref.RefSearch(512, 128, {}, X)
    ids, dists = [], []
    for q in Q:
        s.query_sample = [float(x) for x in q]
        r = s.exact_search_nn(20, include_distances=True)
        ids.append(r[0])
        dists.append(r[1])
    np.savez_compressed(os.path.join(HERE, "exact_search_512x128.npz"),
                        X=X, Q=Q, ids=np.array(ids, dtype=np.int64), dists=np.array(dists, dtype=np.float64))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
