"""Host side of `morna search`: the counterpart of MornaSearch.

Mirrors commanderson/morna morna.py:522-787: same constructor (loads the index
file set written by MornaIndex.save), same `update_query` / `finalize_query` /
`search_nn` / `exact_search_nn` / `search_member_n` signatures and return
shapes, same errors.  Query construction (a dict of summed coverages, then one
hash + idf per distinct junction) stays in Python exactly as in the reference;
the searches run in libmorna_hip.so.

`meta_db=True` appends the keywords of `<basename>.meta.mor` to the results
(morna.py:666-676; metadb.py).
"""
import os
import pickle
import sys
from collections import defaultdict
from math import log

import numpy as np

from . import _lib
from .annoy import AnnoyIndex
from .metadb import lookup_meta


def results_output(results, out=None):
    """Human-readable result lines (morna.py:116-127)."""
    out = out or sys.stdout
    for i in range(len(results[0])):
        out.write(str(i + 1) + ".")
        for lst in results:
            out.write("\t" + str(lst[i]))
        out.write("\n")


class MornaSearch(object):
    def __init__(self, basename, device=0, rank=None, world=None):
        """rank / world: one process per shard of an index built with --shards (shards.DistShards); None: this process
        holds the whole index (or all of its shards, shards.LocalShards)."""
        self.basename = basename
        with open(basename + ".stats.mor") as stats_stream:
            self.sample_count = int(stats_stream.readline())
            self.index_size = int(stats_stream.readline())
            self.dim = int(stats_stream.readline())
        self.query = defaultdict(int)
        self.query_sample = [0.0 for _ in range(self.dim)]
        if world is not None and world > 1:                    # torchrun: this process serves shard `rank`
            from .shards import DistShards
            self.annoy_index = DistShards(basename, self.dim, rank, world, device=device)
        elif os.path.exists(basename + ".shards.mor"):         # written by `morna index --shards G` (index.py)
            from .shards import LocalShards
            self.annoy_index = LocalShards(basename, self.dim, device=device)
        else:
            self.annoy_index = AnnoyIndex(self.dim, metric="angular", device=device)
            self.annoy_index.load(basename + '.annoy.mor')
        with open(basename + ".freq.mor", "rb") as pickle_stream:
            self.sample_frequencies = defaultdict(int, pickle.load(pickle_stream))
        with open(basename + ".map.mor", "rb") as pickle_stream:
            self.internal_id_map = pickle.load(pickle_stream)

    def inverse_lookup(self, internal_id):
        """External sample id that owns `internal_id`, None when nobody does (contract of morna.py:552-572:
        the map is one-to-one, a second owner is a RuntimeError)."""
        owners = [sample_id for sample_id, mapped in self.internal_id_map.items() if mapped == internal_id]
        if len(owners) > 1:
            raise RuntimeError(str(internal_id) + " does not have unique mapping in self.internal_id_map.")
        return owners[0] if owners else None

    def update_query(self, junction):
        """Sum the coverage of one (chrom, start, end, coverage, ...) junction (morna.py:597-607)."""
        self.query[tuple(junction[:3])] += int(junction[3])

    def finalize_query(self):
        """query_sample from the summed coverages (arithmetic contract of morna.py:609-629).

        One term per distinct junction, in the dict's insertion order: weight = log(sample_count / df) with the
        index's FINAL document frequency of the key "chrom start end" (0 for a junction the index never saw),
        term = coverage * weight, added with the sign of the key's 32-bit hash into column hash mod dim (Python's
        floored modulo), all in fp64 -- the terms a column receives are added in that order.
        """
        hash32 = _lib.lib().morna_hash32
        dense = [0.0] * self.dim
        for parts, coverage in self.query.items():
            text = ' '.join(map(str, parts))
            df = self.sample_frequencies.get(text, 0)
            weight = log(float(self.sample_count) / df) if df else 0
            raw = text.encode("ascii")
            h = int(hash32(raw, len(raw)))          # mmh3.hash: signed
            term = coverage * weight
            dense[h % self.dim] += -term if h < 0 else term
        self.query_sample = dense

    def _with_meta(self, results, meta_db):
        """Append the metadata keywords of each result (morna.py:666-676)."""
        if not meta_db:
            return results
        sample_ids = [self.inverse_lookup(internal_id) for internal_id in results[0]]
        return results + (lookup_meta(self.basename, sample_ids),)

    def search_nn(self, num_neighbors, search_k, include_distances=True, meta_db=False):
        """Approximate neighbours of query_sample (morna.py:632-678)."""
        if include_distances:
            results = self.annoy_index.get_nns_by_vector([feature for feature in self.query_sample],
                                                         num_neighbors, search_k, include_distances)
        else:
            results = (self.annoy_index.get_nns_by_vector([feature for feature in self.query_sample],
                                                          num_neighbors, search_k, include_distances),)
        return self._with_meta(results, meta_db)

    def exact_search_nn(self, num_neighbors, include_distances=True, meta_db=False):
        """Brute-force neighbours with cosine_distance (morna.py:681-730)."""
        ids, d, cnt = self.annoy_index.exact_search_batch(np.array([self.query_sample], dtype=np.float64),
                                                          num_neighbors)
        m = int(cnt[0])
        if m < 0:
            # some indexed row gives cosine_distance a negative radicand: the reference's math.sqrt raises while it
            # walks the rows (morna.py:101-114, 697-700), whatever that row's rank would have been
            raise ValueError("math domain error")
        results = ([int(x) for x in ids[0, :m]],)
        if include_distances:
            results += ([float(x) for x in d[0, :m]],)
        return self._with_meta(results, meta_db)

    def search_member_n(self, query_id, num_neighbors, search_k, include_distances=True, meta_db=False):
        """Neighbours of an indexed sample (morna.py:733-787)."""
        print("querying by sample id " + str(query_id))
        try:
            internal_id = self.internal_id_map[query_id]
        except KeyError:
            raise ValueError("Querying sample id " + str(query_id)
                             + " is not possible because no internal id is mapped to that "
                             + "sample id. Likely no sample with that id was included "
                             + "in the index.")
        print("this is internal id " + str(internal_id))
        if include_distances:
            results = self.annoy_index.get_nns_by_item(internal_id, num_neighbors, search_k, include_distances)
        else:
            results = (self.annoy_index.get_nns_by_item(internal_id, num_neighbors, search_k, include_distances),)
        return self._with_meta(results, meta_db)
