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
    def __init__(self, basename, device=0):
        self.basename = basename
        with open(basename + ".stats.mor") as stats_stream:
            self.sample_count = int(stats_stream.readline())
            self.index_size = int(stats_stream.readline())
            self.dim = int(stats_stream.readline())
        self.query = defaultdict(int)
        self.query_sample = [0.0 for _ in range(self.dim)]
        self.annoy_index = AnnoyIndex(self.dim, metric="angular", device=device)
        self.annoy_index.load(basename + '.annoy.mor')
        with open(basename + ".freq.mor", "rb") as pickle_stream:
            self.sample_frequencies = defaultdict(int, pickle.load(pickle_stream))
        with open(basename + ".map.mor", "rb") as pickle_stream:
            self.internal_id_map = pickle.load(pickle_stream)

    def inverse_lookup(self, internal_id):
        """sample id whose internal id is `internal_id` (morna.py:552-572)."""
        match = None
        found_one_already = False
        for sample_id in self.internal_id_map.keys():
            if self.internal_id_map[sample_id] == internal_id:
                match = sample_id
                if found_one_already:
                    raise RuntimeError(str(internal_id) + " does not have unique mapping in self.internal_id_map.")
                found_one_already = True
        return match

    def update_query(self, junction):
        """Sum the coverage of one (chrom, start, end, coverage, ...) junction (morna.py:597-607)."""
        self.query[tuple(junction[:3])] += int(junction[3])

    def finalize_query(self):
        """Dense fp64 query vector from the coverage dict (morna.py:609-629)."""
        self.query_sample = [0.0 for _ in range(self.dim)]
        for junction in self.query.keys():
            hashable_junction = ' '.join(str(_) for _ in junction)
            if self.sample_frequencies[hashable_junction] == 0:
                idf_value = 0
            else:
                idf_value = log(float(self.sample_count) / self.sample_frequencies[hashable_junction])
            key = hashable_junction.encode("ascii")
            hash_value = int(_lib.lib().morna_hash32(key, len(key)))       # mmh3.hash
            multiplier = (-1 if hash_value < 0 else 1)
            self.query_sample[hash_value % self.dim] += (multiplier * (self.query[junction] * idf_value))

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
        if m and np.isnan(d[0, :m]).any():
            raise ValueError("math domain error")     # math.sqrt of a negative radicand in the reference
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
