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
        self._save_global(basename)

    def _save_global(self, basename):
        """Everything of the file set but the matrix + forest blob."""
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


class _OwnedArrays(dict):
    """views into memory of a ParsedLines: holds the owner so that they stay valid as long as the dict does"""
    owner = None


class ParsedLines(object):
    """Result of the native pre-pass (morna_parse_intropolis): the kept junction
    lines of an intropolis file as the arrays the C ABI stages, owned by the library."""

    def __init__(self, path, sample_count=None, sample_threshold=100, cache=None, _ptr=None):
        """cache: path of the binary pre-tokenised cache.  It is used when it was written from this very
        file (same size and mtime) with the same sample_count argument and threshold; otherwise the
        file is parsed and the cache (re)written."""
        import ctypes as C
        import os
        from ._lib import check, lib
        self._p = C.c_void_p()
        self.from_cache = False
        if _ptr is not None:                     # lines the library already holds (from_arrays, shard)
            self._p = _ptr
            self._read_counts()
            return
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
        self._read_counts()

    def _read_counts(self):
        import ctypes as C
        from ._lib import check, lib
        counts = np.zeros(8, np.int64)
        check(lib().morna_lines_counts(self._p, counts.ctypes.data_as(C.c_void_p)))
        (self.n_lines, self.nnz, self.n_items, self.skipped, self.sample_count, self.key_bytes_n,
         self.n_keys, self.lines_read) = [int(x) for x in counts]
        info = np.zeros(4, np.int64)
        check(lib().morna_lines_shard_info(self._p, info.ctypes.data_as(C.c_void_p)))
        self.shard_rank, self.shard_world, self.id_offset, self.n_items_global = [int(x) for x in info]

    @classmethod
    def from_arrays(cls, prep, sample_count):
        """Lines tokenised by the caller (prepare_csr's dict, MornaIndex's buffer): key_bytes, key_off, row_ptr, ids,
        cov, idf, ext_ids."""
        import ctypes as C
        from ._lib import check, lib, ptr
        a = {k: np.ascontiguousarray(prep[k], dtype=t) for k, t in (
            ("key_bytes", np.uint8), ("key_off", np.int64), ("row_ptr", np.int64), ("ids", np.int32), ("cov", np.int32),
            ("idf", np.float64), ("ext_ids", np.int64))}
        p = C.c_void_p()
        check(lib().morna_lines_from_arrays(ptr(a["key_bytes"]), ptr(a["key_off"]), len(a["idf"]), ptr(a["row_ptr"]),
                                            ptr(a["ids"]), ptr(a["cov"]), ptr(a["idf"]), ptr(a["ext_ids"]),
                                            len(a["ext_ids"]), int(sample_count), C.byref(p)))
        return cls(None, _ptr=p)

    def shard(self, rank, world):
        """The lines of row shard `rank` of `world` (morna_lines_shard): global idf and first-seen ids, entries of the
        shard's own items only, ids renumbered from 0."""
        import ctypes as C
        from ._lib import check, lib
        p = C.c_void_p()
        check(lib().morna_lines_shard(self._p, int(rank), int(world), C.byref(p)))
        return ParsedLines(None, _ptr=p)

    def __del__(self):
        p = getattr(self, "_p", None)
        if p is not None and p.value:
            try:
                from ._lib import lib
                lib().morna_lines_free(p)
            except Exception:                          # interpreter shutting down
                pass
            self._p = None

    def arrays(self):
        """numpy views of the library-owned arrays.  The dict keeps this object (and with it the arrays) alive."""
        import ctypes as C
        from ._lib import check, lib
        ptrs = [C.c_void_p() for _ in range(7)]
        check(lib().morna_lines_arrays(self._p, *[C.byref(p) for p in ptrs]))

        def view(p, ctype, n):
            if n == 0 or not p.value:
                return np.zeros(0, dtype=np.dtype(ctype))
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n,))
        out = _OwnedArrays(key_bytes=view(ptrs[0], C.c_uint8, self.key_bytes_n),
                           key_off=view(ptrs[1], C.c_int64, self.n_lines + 1),
                           row_ptr=view(ptrs[2], C.c_int64, self.n_lines + 1),
                           ids=view(ptrs[3], C.c_int32, self.nnz), cov=view(ptrs[4], C.c_int32, self.nnz),
                           idf=view(ptrs[5], C.c_double, self.n_lines), ext_ids=view(ptrs[6], C.c_int64, self.n_items))
        out.owner = self
        return out

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


def shard_bounds(n_items, world):
    """Global id range of every row shard: rank g owns [g * ceil(N / G), (g + 1) * ceil(N / G))."""
    per = (int(n_items) + world - 1) // world
    return [min(int(n_items), per * g) for g in range(world + 1)]


def shard_basename(basename, rank, world):
    """File-set prefix of one row shard's matrix + forest blob."""
    return "%s.shard%dof%d" % (basename, rank, world)


def shard_csr(prep, rank, world):
    """prepare_csr's result cut to row shard `rank` of `world`: what morna_lines_shard does, in numpy (the lines keep the
    GLOBAL idf; a line's entries are those of the shard's items, ids from 0; empty lines go)."""
    lo, hi = shard_bounds(prep["n_items"], world)[rank:rank + 2]
    ids = np.asarray(prep["ids"])
    keep = (ids >= lo) & (ids < hi)
    lens = np.add.reduceat(keep.astype(np.int64), prep["row_ptr"][:-1]) if len(prep["idf"]) else np.zeros(0, np.int64)
    lens[np.diff(prep["row_ptr"]) == 0] = 0                     # reduceat returns the next element for an empty slice
    lines = np.nonzero(lens)[0]
    row_ptr = np.zeros(len(lines) + 1, np.int64)
    row_ptr[1:] = np.cumsum(lens[lines])
    klen = np.diff(prep["key_off"])
    key_off = np.zeros(len(lines) + 1, np.int64)
    key_off[1:] = np.cumsum(klen[lines])
    kb = np.asarray(prep["key_bytes"])
    key_bytes = kb[np.repeat(np.isin(np.arange(len(klen)), lines), klen)] if len(lines) else np.zeros(0, np.uint8)
    return dict(key_bytes=key_bytes, key_off=key_off, row_ptr=row_ptr, ids=(ids[keep] - lo).astype(np.int32),
                cov=np.asarray(prep["cov"])[keep], idf=np.asarray(prep["idf"])[lines], n_items=hi - lo,
                ext_ids=np.asarray(prep["ext_ids"])[lo:hi], id_offset=lo, n_items_global=prep["n_items"])


def build_shard(parsed, features, n_trees, rank, world, device=0, seed=0):
    """Row shard `rank` of `world` of ONE parsed data set: its lines (global idf, global first-seen ids), its feature
    matrix and its own forest.  Returns (AnnoyIndex, id_offset, n_local)."""
    part = parsed.shard(rank, world) if world > 1 else parsed
    a = AnnoyIndex(features, metric='angular', device=device)
    if part.n_items == 0:
        raise ValueError("row shard %d of %d is empty: %d samples do not fill %d shards" % (rank, world, parsed.n_items, world))
    part.stage(a)
    a.build_features(part.n_items)
    a.unstage_junctions()
    a.build(n_trees, seed=seed)
    return a, part.id_offset, part.n_items


def go_index_native(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size, verbose,
                    metafile, device=0, save=True, seed=0, cache=None, shards=1, rank=None):
    """go_index with the tokenising loop done by the library (morna_parse_intropolis)
    instead of the Python interpreter; same index, same files.  `cache`: binary
    pre-tokenised cache file to reuse / write (ParsedLines).

    shards > 1: the rows are cut into `shards` contiguous ranges of internal ids, each with its own matrix + forest
    blob (<basename>.shard<g>of<G>.annoy.mor); the global files (.stats / .freq / .map / .shards.mor) describe the
    whole index.  rank=None builds every shard in this process, one after the other; rank=g builds shard g only (one
    process per GPU) and rank 0 writes the global files."""
    parsed = ParsedLines(intropolis, sample_count, sample_threshold, cache=cache)
    if verbose:
        print('\nThere are {} samples.'.format(parsed.sample_count))
    if shards > 1:
        return _go_index_sharded(parsed, basename, features, n_trees, sample_threshold, buffer_size, verbose, metafile,
                                 device, save, seed, shards, rank)
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


def _go_index_sharded(parsed, basename, features, n_trees, sample_threshold, buffer_size, verbose, metafile, device,
                      save, seed, shards, rank):
    if parsed.n_items == 0:
        raise ValueError("No internal ids were assigned, indicating that no samples were added to the index. "
                         "Likely caused when no junctions pass the sample threshold.")
    bounds = shard_bounds(parsed.n_items, shards)
    built = []
    for g in (range(shards) if rank is None else [rank]):
        a, off, n = build_shard(parsed, features, n_trees, g, shards, device=device, seed=seed)
        assert (off, off + n) == (bounds[g], bounds[g + 1])
        if save:
            a.save(shard_basename(basename, g, shards) + '.annoy.mor')
        built.append(a)
    if save and (rank is None or rank == 0):
        # the files that describe the WHOLE index: as MornaIndex.save writes them (morna.py:443-455), ids global
        head = MornaIndex.__new__(MornaIndex)
        head.sample_count, head.new_internal_id, head.dim, head.metafile = parsed.sample_count, parsed.n_items, features, metafile
        head.sample_frequencies = parsed.frequencies()
        head.internal_id_map = {int(s): i for i, s in enumerate(parsed.arrays()["ext_ids"].tolist())}
        head._save_global(basename)
        with open(basename + ".shards.mor", "w") as fh:
            fh.write(str(shards) + "\n" + " ".join(str(b) for b in bounds) + "\n")
    return built if rank is None else built[0]


def go_index(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size, verbose,
             metafile, device=0, save=True, seed=0, native=False, cache=None, shards=1, rank=None):
    """`morna index` (morna.py:824-865): gzipped intropolis file -> index files."""
    if native or cache or shards > 1:
        return go_index_native(intropolis, basename, features, n_trees, sample_count, sample_threshold, buffer_size,
                               verbose, metafile, device=device, save=save, seed=seed, cache=cache, shards=shards,
                               rank=rank)
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
    # first-seen order of sample ids (morna.py:377-382)
    if len(s_kept) and s_kept.dtype.kind in "iu" and int(s_kept.min()) >= 0 and int(s_kept.max()) < (1 << 27):
        # small non-negative ids: a table indexed by the id.  Assigning the positions in REVERSE leaves each id's first
        # position in its slot (numpy stores repeated indices in order, the last store wins)
        first_at = np.full(int(s_kept.max()) + 1, -1, np.int64)
        first_at[s_kept[::-1]] = np.arange(len(s_kept) - 1, -1, -1, dtype=np.int64)
        uniq = np.nonzero(first_at >= 0)[0].astype(s_kept.dtype)
        order = np.argsort(first_at[uniq], kind="stable")
        ext_ids = uniq[order]
        table = np.zeros(len(first_at), np.int32)
        table[ext_ids] = np.arange(len(ext_ids), dtype=np.int32)
        ids = table[s_kept]
    else:
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
