"""Pure-Python (py3) restatement of morna.py's hot-path functions.

TEST INFRASTRUCTURE: see oracle/__init__.py for who may import this.

The reference (/root/reference/morna.py) is Python-2-only and its imports
(annoy, mmh3, BitVector) are not installed, so it cannot be imported here; this
module restates, function by function, what the reference computes, on the same
Python data structures (dict of lists of float, first-seen internal ids), for
inputs small enough for interpreter loops.  Each function cites the lines it
follows.  It is the generator of tests/golden/*.json
(tests/golden/make_golden.py) and is itself pinned by the reference's embedded
known answers (morna.py:1176-1187, 1267-1278, 1312-1323).
"""
import bisect
import gzip
import struct
from collections import defaultdict
from math import log, sqrt

import numpy as np


def mmh3_hash(key, seed=0):
    """mmh3.hash(key): MurmurHash3_x86_32, signed (morna.py:369, 591, 625)."""
    data = key.encode("ascii") if isinstance(key, str) else bytes(key)
    c1, c2 = 0xcc9e2d51, 0x1b873593
    h = seed & 0xFFFFFFFF
    n = len(data)
    for i in range(0, n - n % 4, 4):
        k = struct.unpack_from("<I", data, i)[0]
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xe6546b64) & 0xFFFFFFFF
    tail = data[n - n % 4:]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
    h ^= n
    h ^= h >> 16
    h = (h * 0x85ebca6b) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & 0xFFFFFFFF
    h ^= h >> 16
    return h - (1 << 32) if h & 0x80000000 else h


def count_samples(lines):
    """count_samples (morna.py:789-822): distinct sample-id STRINGS of column -2."""
    samples = set()
    for line in lines:
        samples.update(line.split('\t')[-2].split(','))
    return len(samples)


def tokenize_line(line):
    """go_index's per-line parse (morna.py:848-853)."""
    tokens = line.strip().split('\t')
    return (' '.join(tokens[:3]),
            [int(s) for s in tokens[-2].split(',')],
            [int(c) for c in tokens[-1].split(',')])


class RefIndex(object):
    """MornaIndex without the junction database (morna.py:146-218, 344-425)."""

    def __init__(self, sample_count, dim=3000, sample_threshold=100):
        self.sample_count = sample_count
        self.dim = dim
        self.sample_threshold = sample_threshold
        self.internal_id_map = {}
        self.new_internal_id = 0
        self.sample_frequencies = defaultdict(int)
        self.sample_feature_matrix = defaultdict(lambda: [0.0 for _ in range(dim)])
        self.skipped = 0
        self.junc_id = -1

    def add_junction(self, junction, samples, coverages):
        """morna.py:344-388 (update_junction_dbs at 359 is out of scope)."""
        self.junc_id += 1
        if len(samples) < self.sample_threshold:
            self.skipped += 1
            return
        self.sample_frequencies[junction] += len(samples)
        hashed_value = mmh3_hash(junction)
        multiplier = (-1 if hashed_value < 0 else 1)
        hashed_value = hashed_value % self.dim
        idf_value = log(float(self.sample_count) / self.sample_frequencies[junction])
        for sample_id, coverage in zip(samples, coverages):
            if sample_id not in self.internal_id_map:
                self.internal_id_map[sample_id] = self.new_internal_id
                self.new_internal_id += 1
            internal_id = self.internal_id_map[sample_id]
            tf_idf_score = (coverage * idf_value)
            self.sample_feature_matrix[int(internal_id)][hashed_value] += (multiplier * tf_idf_score)

    def matrix64(self):
        """Dense fp64 matrix by internal id; build() raises when empty (morna.py:399-403)."""
        if self.new_internal_id == 0:
            raise ValueError("No internal ids were assigned")
        M = np.zeros((self.new_internal_id, self.dim), dtype=np.float64)
        for i, row in self.sample_feature_matrix.items():
            M[i] = row
        return M

    def matrix32(self):
        """What AnnoyIndex.add_item stores: fp64 -> fp32 (morna.py:405-424)."""
        return self.matrix64().astype(np.float32)


def go_index_lines(lines, features, sample_count=None, sample_threshold=100):
    """go_index (morna.py:824-865) on an iterable of text lines."""
    lines = list(lines)
    if not sample_count:
        sample_count = count_samples(lines)
    idx = RefIndex(sample_count, dim=features, sample_threshold=sample_threshold)
    for line in lines:
        idx.add_junction(*tokenize_line(line))
    return idx


def go_index_gz(path, features, sample_count=None, sample_threshold=100):
    with gzip.open(path, "rt") as fh:
        return go_index_lines(fh.readlines(), features, sample_count, sample_threshold)


class RefSearch(object):
    """MornaSearch query construction and exact search (morna.py:522-730)."""

    def __init__(self, sample_count, dim, sample_frequencies, X32):
        self.sample_count = sample_count
        self.dim = dim
        self.sample_frequencies = defaultdict(int, sample_frequencies)
        self.X = np.asarray(X32, dtype=np.float32)
        self.index_size = self.X.shape[0]
        self.query = defaultdict(int)
        self.query_sample = [0.0 for _ in range(dim)]

    def update_query(self, junction):
        """morna.py:597-607"""
        self.query[tuple(junction[:3])] += int(junction[3])

    def finalize_query(self):
        """morna.py:609-629"""
        self.query_sample = [0.0 for _ in range(self.dim)]
        for junction in self.query.keys():
            hashable_junction = ' '.join(str(_) for _ in junction)
            if self.sample_frequencies[hashable_junction] == 0:
                idf_value = 0
            else:
                idf_value = log(float(self.sample_count) / self.sample_frequencies[hashable_junction])
            hash_value = mmh3_hash(hashable_junction)
            multiplier = (-1 if hash_value < 0 else 1)
            self.query_sample[hash_value % self.dim] += (multiplier * (self.query[junction] * idf_value))

    def exact_search_nn(self, num_neighbors, include_distances=True):
        """morna.py:681-716"""
        neighbor_indexes = []
        neighbor_distances = []
        for i in range(0, self.index_size):
            current_distance = cosine_distance([float(x) for x in self.X[i]], self.query_sample)
            insert_point = bisect.bisect_left(neighbor_distances, current_distance)
            if insert_point < num_neighbors:
                neighbor_distances.insert(insert_point, current_distance)
                neighbor_indexes.insert(insert_point, i)
            if len(neighbor_distances) > num_neighbors:
                neighbor_distances = neighbor_distances[0:num_neighbors]
                neighbor_indexes = neighbor_indexes[0:num_neighbors]
        results = (neighbor_indexes,)
        if include_distances:
            results += (neighbor_distances,)
        return results


def cosine_distance(v1, v2):
    """morna.py:101-114"""
    pp = 0.0
    qq = 0.0
    pq = 0.0
    for (i, j) in zip(v1, v2):
        pp += i * i
        qq += j * j
        pq += i * j
    ppqq = pp * qq
    if ppqq > 0.0:
        distance = 2.0 - 2.0 * pq / sqrt(ppqq)
    else:
        distance = 2.0
    return sqrt(distance)


def angular_order_exact(X32, item, n):
    """What annoy returns from get_nns_by_item(item, n, search_k) when every
    root is a single leaf (N <= K = f + 2, the case of morna.py:1189-1193):
    all items sorted by (angular distance, id).  Distances are evaluated in
    fp64 here; groups of mathematically tied rows are returned together so the
    caller can compare tie-aware.  Returns (order, dist)."""
    X = np.asarray(X32, dtype=np.float64)
    v = X[item]
    pp = float(v @ v)
    d = np.empty(X.shape[0])
    for j in range(X.shape[0]):
        qq = float(X[j] @ X[j])
        pq = float(v @ X[j])
        d[j] = 2.0 - 2.0 * pq / sqrt(pp * qq) if pp * qq > 0 else 2.0
    order = sorted(range(X.shape[0]), key=lambda j: (round(d[j], 9), j))
    return order[:n], d


def results_lines(results):
    """results_output (morna.py:116-127) as a list of strings."""
    out = []
    for i in range(len(results[0])):
        s = str(i + 1) + "."
        for lst in results:
            s += "\t" + str(lst[i])
        out.append(s + "\n")
    return out
