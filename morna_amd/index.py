"""Host side of `morna index`: the counterpart of MornaIndex / go_index.

Mirrors commanderson/morna morna.py:146-520 (MornaIndex), 789-822
(count_samples) and 824-865 (go_index) for the hot path: same class and method
names, same arguments, same errors.  What stays on the host is exactly what is
sequential and string-shaped in the reference -- tokenising, the sample
threshold, the cumulative frequency dict, first-seen internal ids and
idf = log(sample_count / freq) (libm, so it is bit-identical to Python's).
Everything per-nnz and everything numeric runs in libmorna_hip.so: junction
lines are buffered as CSR arrays and handed over once by build().

Out of scope here: the junctions-by-sample
sqlite shards (update_junction_dbs).
"""
import gzip
import pickle
import sys
from collections import defaultdict
from math import log

import numpy as np

from .annoy import AnnoyIndex
from .metadb import write_meta_db


def count_samples(introp_file_handle, verbose=False):
    """Distinct sample-id strings of column -2 (morna.py:789-822)."""
    samples = set()
    for i, line in enumerate(introp_file_handle):
        if verbose and i % 100 == 0:
            sys.stdout.write(str(i) + " lines into sample count, " + str(len(samples)) + " samples so far.\r")
            sys.stdout.flush()
        if isinstance(line, bytes):
            line = line.decode("ascii")
        samples.update(line.split('\t')[-2].split(','))
    return len(samples)


class JunctionBuffer(object):
    """Kept junction lines in file order, as the CSR arrays the C ABI takes."""

    def __init__(self):
        self.keys = []          # bytes per kept line
        self.row_len = []
        self.ids = []           # list of int32 arrays (internal ids)
        self.cov = []           # list of int32 arrays
        self.idf = []

    def append(self, key_bytes, ids, cov, idf):
        self.keys.append(key_bytes)
        self.row_len.append(len(ids))
        self.ids.append(ids)
        self.cov.append(cov)
        self.idf.append(idf)

    def arrays(self):
        J = len(self.keys)
        key_off = np.zeros(J + 1, np.int64)
        row_ptr = np.zeros(J + 1, np.int64)
        if J:
            key_off[1:] = np.cumsum([len(k) for k in self.keys])
            row_ptr[1:] = np.cumsum(self.row_len)
        key_bytes = np.frombuffer(b"".join(self.keys), dtype=np.uint8) if J else np.zeros(0, np.uint8)
        ids = np.concatenate(self.ids).astype(np.int32, copy=False) if J else np.zeros(0, np.int32)
        cov = np.concatenate(self.cov).astype(np.int32, copy=False) if J else np.zeros(0, np.int32)
        idf = np.array(self.idf, dtype=np.float64)
        return key_bytes, key_off, row_ptr, ids, cov, idf


class MornaIndex(AnnoyIndex):
    """AnnoyIndex augmented with morna's parameters (morna.py:146-218)."""

    def __init__(self, sample_count, basename, dim=3000, sample_threshold=100, metafile=None,
                 buffer_size=1024, device=0):
        super(MornaIndex, self).__init__(dim, metric='angular', device=device)
        self.sample_count = sample_count
        self.basename = basename
        self.internal_id_map = {}
        self.new_internal_id = 0
        self.sample_frequencies = defaultdict(int)
        self.metafile = metafile
        self.dim = self.dimension_count = dim
        self.sample_threshold = sample_threshold
        self.skipped = 0
        self.buffer_size = buffer_size
        self.junc_id = -1
        self._lines = JunctionBuffer()

    def add_junction(self, junction, samples, coverages):
        """Buffers the contribution of one junction line (morna.py:344-388).

        The threshold test, the cumulative frequency, first-seen internal ids and
        the idf are computed here exactly as the reference does; the hash, the
        sign/column and the per-sample accumulation are done on the GPU by build().
        """
        self.junc_id += 1
        if len(samples) < self.sample_threshold:
            self.skipped += 1
            return
        self.sample_frequencies[junction] += len(samples)
        idf_value = log(float(self.sample_count) / self.sample_frequencies[junction])
        # the reference walks zip(samples, coverages): a longer list is cut to the shorter one, and only the
        # samples it reaches get internal ids (the threshold and the frequency above use len(samples))
        n = min(len(samples), len(coverages))
        ids = np.empty(n, np.int32)
        id_map = self.internal_id_map
        for t in range(n):
            sample_id = samples[t]
            internal_id = id_map.get(sample_id)
            if internal_id is None:
                internal_id = id_map[sample_id] = self.new_internal_id
                self.new_internal_id += 1
            ids[t] = internal_id
        cov = np.asarray(coverages[:n], dtype=np.int64)
        if n and (cov.max() > 2**31 - 1 or cov.min() < -2**31):
            raise OverflowError("coverage of junction %r does not fit 32 bits (the staged arrays are int32)" % (junction,))
        self._lines.append(junction.encode("ascii"), ids, cov.astype(np.int32), idf_value)

    def build(self, n_trees, verbose=False, seed=0):
        """Feature matrix on the GPU, then the forest (morna.py:390-425)."""
        if self.new_internal_id == 0:
            raise ValueError("No internal ids were assigned, indicating that "
                             + "no samples were added to the index. Likely "
                             + "caused when no junctions pass the sample "
                             + "threshold.")
        self.stage_junctions(*self._lines.arrays())
        # intropolis lines list their samples by ascending sample id: that order lets the GPU read each (sample,
        # coverage) pair once (a hint: the matrix is the same without it)
        by_internal = sorted(self.internal_id_map, key=self.internal_id_map.get)
        try:
            self.stage_item_order(np.asarray(by_internal, dtype=np.int64))
        except (TypeError, ValueError, OverflowError):
            pass                                        # sample ids that are not integers: no hint
        self.build_features(self.new_internal_id)
        self.unstage_junctions()
        if verbose:
            sys.stderr.write('\nAdded a total of {} samples to Annoy index.\n'.format(self.new_internal_id))
            sys.stderr.write('{} junctions skipped for not meeting sample threshold\n'.format(self.skipped))
        super(MornaIndex, self).build(n_trees, seed=seed)

    def save(self, basename):
        """Index file set of the reference (morna.py:427-455), minus the sqlite shards."""
        super(MornaIndex, self).save(basename + '.annoy.mor')
        with open(basename + ".stats.mor", 'w') as stats_stream:
            stats_stream.write(str(self.sample_count) + "\n")
            stats_stream.write(str(self.new_internal_id) + "\n")
            stats_stream.write(str(self.dim) + "\n")
        with open(basename + ".freq.mor", 'wb') as pickle_stream:
            pickle.dump(dict(self.sample_frequencies), pickle_stream, 2)
        with open(basename + ".map.mor", 'wb') as pickle_stream:
            pickle.dump(self.internal_id_map, pickle_stream, 2)
        if self.metafile:
            write_meta_db(self.metafile, basename)      # morna.py:494-520


def tokenize_line(line):
    """go_index's per-line parse (morna.py:848-853)."""
    tokens = line.strip().split('\t')
    return (' '.join(tokens[:3]),
            [int(s) for s in tokens[-2].split(',')],
            [int(c) for c in tokens[-1].split(',')])


class ParsedLines(object):
    """Result of the native pre-pass (morna_parse_intropolis): the kept junction
    lines of an intropolis file as the arrays the C ABI stages, owned by the library."""

    def __init__(self, path, sample_count=None, sample_threshold=100, cache=None):
        """cache: path of the binary pre-tokenised cache.  It is used when it was written from this very
        file (same size and mtime) with the same sample_count argument and threshold; otherwise the
        file is parsed and the cache (re)written."""
        import ctypes as C
        import os
        from ._lib import check, lib
        self._p = C.c_void_p()
        self.from_cache = False
        tag = None
        if cache:
            st = os.stat(path)
            tag = np.array([st.st_size, st.st_mtime_ns, int(sample_count or 0), int(sample_threshold)], np.int64)
            if os.path.exists(cache):
                got = np.zeros(4, np.int64)
                rc = lib().morna_lines_load(str(cache).encode(), got.ctypes.data_as(C.c_void_p), C.byref(self._p))
                if rc == 0 and (got == tag).all():
                    self.from_cache = True
                elif rc == 0:                                  # a cache of some other file or other parameters
                    lib().morna_lines_free(self._p)
                    self._p = C.c_void_p()
        if not self.from_cache:
            check(lib().morna_parse_intropolis(str(path).encode(), int(sample_count or 0), int(sample_threshold),
                                               C.byref(self._p)))
            if cache:
                check(lib().morna_lines_save(self._p, str(cache).encode(), tag.ctypes.data_as(C.c_void_p)))
        counts = np.zeros(8, np.int64)
        check(lib().morna_lines_counts(self._p, counts.ctypes.data_as(C.c_void_p)))
        (self.n_lines, self.nnz, self.n_items, self.skipped, self.sample_count, self.key_bytes_n,
         self.n_keys, self.lines_read) = [int(x) for x in counts]

    def __del__(self):
        p = getattr(self, "_p", None)
        if p is not None and p.value:
            from ._lib import lib
            lib().morna_lines_free(p)
            self._p = None

    def arrays(self):
        """numpy views of the library-owned arrays (valid while this object lives)."""
        import ctypes as C
        from ._lib import check, lib
        ptrs = [C.c_void_p() for _ in range(7)]
        check(lib().morna_lines_arrays(self._p, *[C.byref(p) for p in ptrs]))

        def view(p, ctype, n):
            if n == 0 or not p.value:
                return np.zeros(0, dtype=np.dtype(ctype))
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n,))
        return dict(key_bytes=view(ptrs[0], C.c_uint8, self.key_bytes_n),
                    key_off=view(ptrs[1], C.c_int64, self.n_lines + 1),
                    row_ptr=view(ptrs[2], C.c_int64, self.n_lines + 1),
                    ids=view(ptrs[3], C.c_int32, self.nnz), cov=view(ptrs[4], C.c_int32, self.nnz),
                    idf=view(ptrs[5], C.c_double, self.n_lines), ext_ids=view(ptrs[6], C.c_int64, self.n_items))

    def frequencies(self):
        """junction -> cumulative sample frequency (MornaIndex.sample_frequencies)."""
        import ctypes as C
        from ._lib import check, lib
        out = {}
        key, klen, freq = C.c_char_p(), C.c_int64(), C.c_int64()
        for i in range(self.n_keys):
            check(lib().morna_lines_freq_entry(self._p, i, C.byref(key), C.byref(klen), C.byref(freq)))
            out[C.string_at(key, klen.value).decode("ascii")] = int(freq.value)
        return out

    def stage(self, index):
        from ._lib import check, lib
        check(lib().morna_stage_lines(index._h, self._p))


def go_index_native(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size, verbose,
                    metafile, device=0, save=True, seed=0, cache=None):
    """go_index with the tokenising loop done by the library (morna_parse_intropolis)
    instead of the Python interpreter; same index, same files.  `cache`: binary
    pre-tokenised cache file to reuse / write (ParsedLines)."""
    parsed = ParsedLines(intropolis, sample_count, sample_threshold, cache=cache)
    if verbose:
        print('\nThere are {} samples.'.format(parsed.sample_count))
    morna_index = MornaIndex(parsed.sample_count, basename, dim=features, sample_threshold=sample_threshold,
                             metafile=metafile, buffer_size=buffer_size, device=device)
    morna_index.junc_id = parsed.lines_read - 1
    morna_index.skipped = parsed.skipped
    morna_index.new_internal_id = parsed.n_items
    ext = parsed.arrays()["ext_ids"]
    morna_index.internal_id_map = {int(s): i for i, s in enumerate(ext.tolist())}
    morna_index.sample_frequencies = defaultdict(int, parsed.frequencies())
    if parsed.n_items == 0:
        raise ValueError("No internal ids were assigned, indicating that no samples were added to the index. "
                         "Likely caused when no junctions pass the sample threshold.")
    parsed.stage(morna_index)
    morna_index.build_features(parsed.n_items)
    morna_index.unstage_junctions()
    AnnoyIndex.build(morna_index, n_trees, seed=seed)
    if save:
        morna_index.save(basename)
    return morna_index


def go_index(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size, verbose,
             metafile, device=0, save=True, seed=0, native=False, cache=None):
    """`morna index` (morna.py:824-865): gzipped intropolis file -> index files."""
    if native or cache:
        return go_index_native(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size,
                               verbose, metafile, device=device, save=save, seed=seed, cache=cache)
    if not sample_count:
        with gzip.open(intropolis, "rt") as introp_file_handle:
            sample_count = count_samples(introp_file_handle, verbose)
    if verbose:
        print('\nThere are {} samples.'.format(sample_count))
    morna_index = MornaIndex(sample_count, basename, dim=features, sample_threshold=sample_threshold,
                             metafile=metafile, buffer_size=buffer_size, device=device)
    with gzip.open(intropolis, "rt") as introp_file_handle:
        for i, line in enumerate(introp_file_handle):
            if verbose and i % 1000 == 0:
                sys.stdout.write(str(i) + " lines into index making\r")
                sys.stdout.flush()
            morna_index.add_junction(*tokenize_line(line))
    if verbose:
        print('Finished making index; now building')
    morna_index.build(n_trees, verbose=verbose, seed=seed)
    if save:
        morna_index.save(basename)
    return morna_index


def prepare_csr(keys, row_ptr, samples, cov, sample_count, sample_threshold):
    """Vectorised host pass of add_junction over pre-tokenised lines: threshold,
    cumulative frequency per key, first-seen internal ids, idf.  Same results as
    feeding the lines to MornaIndex.add_junction one by one.

    keys: list of str/bytes (J); row_ptr int64[J+1]; samples/cov int arrays [nnz].
    Returns dict(key_bytes, key_off, row_ptr, ids, cov, idf, n_items, ext_ids,
    freq (dict), skipped).
    """
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    samples = np.asarray(samples)
    cov = np.asarray(cov)
    J = len(keys)
    lens = np.diff(row_ptr)
    keep = lens >= sample_threshold
    kept = np.nonzero(keep)[0]
    freq = defaultdict(int)
    idf = np.empty(len(kept), np.float64)
    kb = []
    for a, j in enumerate(kept):
        k = keys[j]
        if isinstance(k, str):
            k = k.encode("ascii")
        freq[k] += int(lens[j])
        idf[a] = log(float(sample_count) / freq[k])
        kb.append(k)
    # nnz of kept lines, file order
    if len(kept) == J:
        s_kept = samples[row_ptr[0]:row_ptr[-1]]
        c_kept = cov[row_ptr[0]:row_ptr[-1]].astype(np.int32)
    else:
        mask = np.repeat(keep, lens)
        s_kept = samples[row_ptr[0]:row_ptr[-1]][mask]
        c_kept = cov[row_ptr[0]:row_ptr[-1]][mask].astype(np.int32)
    # first-seen order of sample ids
    uniq, first = np.unique(s_kept, return_index=True)
    order = np.argsort(first, kind="stable")
    ext_ids = uniq[order]
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    ids = rank[np.searchsorted(uniq, s_kept)].astype(np.int32)
    new_row_ptr = np.zeros(len(kept) + 1, np.int64)
    new_row_ptr[1:] = np.cumsum(lens[kept])
    key_off = np.zeros(len(kept) + 1, np.int64)
    if kb:
        key_off[1:] = np.cumsum([len(k) for k in kb])
    key_bytes = np.frombuffer(b"".join(kb), dtype=np.uint8) if kb else np.zeros(0, np.uint8)
    return dict(key_bytes=key_bytes, key_off=key_off, row_ptr=new_row_ptr, ids=ids, cov=c_kept, idf=idf,
                n_items=int(len(uniq)), ext_ids=ext_ids, freq={k.decode("ascii"): v for k, v in freq.items()},
                skipped=int(J - len(kept)))
