"""Row shards of ONE data set: the host half -- morna_lines_shard (native) and shard_csr (numpy) -- without
a GPU.  What the cut must preserve is stated by the reference's scatter loop (morna.py:357-388): a cell of the matrix is
the sum, in file order, of mult * (cov * idf) over the lines of its column that list its sample, with idf from the GLOBAL
sample count and cumulative frequency; internal ids are first-seen over the WHOLE file."""
import gzip
import os
from math import log

import numpy as np
import pytest

from morna_amd import index as mindex
from morna_amd.synth import synthetic_intropolis
from oracle import capi

KEYS = ("key_bytes", "key_off", "row_ptr", "ids", "cov", "idf", "ext_ids")


def _small():
    d = synthetic_intropolis(1200, J=900)
    prep = mindex.prepare_csr(d["keys"], d["row_ptr"], d["samples"], d["cov"], d["sample_count"], 60)
    return d, prep


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_native_shard_equals_numpy_shard_and_partitions_the_entries(world):
    d, prep = _small()
    L = mindex.ParsedLines.from_arrays(prep, d["sample_count"])
    assert (L.shard_rank, L.shard_world, L.id_offset, L.n_items_global) == (0, 1, 0, prep["n_items"])
    bounds = mindex.shard_bounds(prep["n_items"], world)
    assert bounds[0] == 0 and bounds[-1] == prep["n_items"]
    seen = 0
    line_of = np.repeat(np.arange(len(prep["idf"])), np.diff(prep["row_ptr"]))
    for g in range(world):
        S = L.shard(g, world)
        a, ref = S.arrays(), mindex.shard_csr(prep, g, world)
        for k in KEYS:
            assert np.array_equal(a[k], ref[k]), (world, g, k)
        assert (S.shard_rank, S.shard_world, S.id_offset, S.n_items_global) == (g, world, bounds[g], prep["n_items"])
        assert S.n_items == bounds[g + 1] - bounds[g] and S.sample_count == d["sample_count"]
        # the shard's entries are exactly the whole's entries of its id range, in file order, with the line's global idf
        mine = (prep["ids"] >= bounds[g]) & (prep["ids"] < bounds[g + 1])
        assert np.array_equal(a["ids"] + bounds[g], prep["ids"][mine])
        assert np.array_equal(a["cov"], prep["cov"][mine])
        kept_lines = np.unique(line_of[mine])
        assert np.array_equal(a["idf"], prep["idf"][kept_lines])
        seen += S.nnz
    assert seen == len(prep["ids"])


def test_shard_rows_of_the_matrix_equal_the_whole_matrix_rows():
    """The claim the sharded build rests on, checked in the reference's own arithmetic on the host: accumulate every
    shard's lines as add_junction does (fp64, file order, mult * (cov * idf)) -- the stacked results are the oracle's
    matrix of the WHOLE data set, bit for bit."""
    d, prep = _small()
    D = 64
    buf, off = capi.pack_keys(d["keys"])
    whole = capi.index_features(buf, off, d["row_ptr"], d["samples"], d["cov"], d["sample_count"], 60, D,
                                max_items=prep["n_items"])
    assert whole["ext_ids"].tolist() == prep["ext_ids"].tolist()
    stacked = []
    for g in range(3):
        sh = mindex.shard_csr(prep, g, 3)
        _h, col, sign = capi.hash_col_sign(np.ascontiguousarray(sh["key_bytes"]) if len(sh["key_bytes"]) else np.zeros(1, np.uint8),
                                           sh["key_off"], D)
        M = np.zeros((sh["n_items"], D), np.float64)
        for j in range(len(sh["idf"])):
            lo, hi = sh["row_ptr"][j], sh["row_ptr"][j + 1]
            # morna.py:386-388: one rounded product cov * idf, one by the sign, one addition per entry; a sample is
            # listed once per line here, so the line's entries touch distinct cells
            M[sh["ids"][lo:hi], col[j]] += float(sign[j]) * (sh["cov"][lo:hi].astype(np.float64) * sh["idf"][j])
        stacked.append(M)
    M = np.concatenate(stacked)
    assert M.tobytes() == whole["M"].tobytes()
    assert M.astype(np.float32).tobytes() == whole["X"].tobytes()


def test_shard_uses_global_frequency_not_the_shards_own():
    keys = ["chr1 1 2", "chr1 5 9", "chr1 1 2"]
    rp = np.array([0, 3, 5, 9])
    s = np.array([7, 9, 4, 9, 3, 7, 3, 11, 4])
    c = np.arange(1, 10)
    prep = mindex.prepare_csr(keys, rp, s, c, 20, 1)
    assert prep["ext_ids"].tolist() == [7, 9, 4, 3, 11]
    L = mindex.ParsedLines.from_arrays(prep, 20)
    s0, s1 = L.shard(0, 2).arrays(), L.shard(1, 2).arrays()              # ids 0..2 | 3..4
    assert s0["idf"].tolist() == [log(20.0 / 3), log(20.0 / 2), log(20.0 / 7)]    # every line keeps an entry in shard 0
    assert s1["idf"].tolist() == [log(20.0 / 2), log(20.0 / 7)]                   # line 0 has none of samples 3, 11
    assert s0["ids"].tolist() == [0, 1, 2, 1, 0, 2] and s0["cov"].tolist() == [1, 2, 3, 4, 6, 9]
    assert s1["ids"].tolist() == [0, 0, 1] and s1["cov"].tolist() == [5, 7, 8]
    assert bytes(s1["key_bytes"]) == b"chr1 5 9chr1 1 2" and s1["key_off"].tolist() == [0, 8, 16]
    assert s0["ext_ids"].tolist() == [7, 9, 4] and s1["ext_ids"].tolist() == [3, 11]


def test_shard_edges():
    d, prep = _small()
    L = mindex.ParsedLines.from_arrays(prep, d["sample_count"])
    for bad in ((-1, 2), (2, 2), (0, 0)):
        with pytest.raises(ValueError):
            L.shard(*bad)
    with pytest.raises(RuntimeError):                  # a shard is not cut again
        L.shard(0, 2).shard(0, 2)
    # more shards than items: the last ones are empty, nothing is lost
    tiny = mindex.prepare_csr(["chr1 1 2"], np.array([0, 3]), np.array([5, 6, 7]), np.ones(3, np.int64), 3, 1)
    T = mindex.ParsedLines.from_arrays(tiny, 3)
    sizes = [T.shard(g, 5).n_items for g in range(5)]
    assert sizes == [1, 1, 1, 0, 0] and T.shard(4, 5).n_lines == 0
    with pytest.raises(IndexError):                    # an id outside [0, n_items)
        broken = dict(tiny)
        broken["ids"] = np.array([0, 1, 3], np.int32)
        mindex.ParsedLines.from_arrays(broken, 3)


def test_parsed_file_then_shards(tmp_path):
    """morna_parse_intropolis -> morna_lines_shard: the shard of a parsed FILE carries the file's frequency table and
    external ids, and equals the cut of the Python pre-pass."""
    d = synthetic_intropolis(800, J=500)
    path = str(tmp_path / "i.tsv.gz")
    with gzip.open(path, "wt") as fh:
        for j, k in enumerate(d["keys"]):
            lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
            fh.write("\t".join(k.split(" ") + ["+", "GT", "AG", ",".join(map(str, d["samples"][lo:hi])),
                                               ",".join(map(str, d["cov"][lo:hi]))]) + "\n")
    P = mindex.ParsedLines(path, None, 40)
    prep = mindex.prepare_csr(d["keys"], d["row_ptr"], d["samples"], d["cov"], P.sample_count, 40)
    for g in range(3):
        S = P.shard(g, 3)
        a, ref = S.arrays(), mindex.shard_csr(prep, g, 3)
        for k in KEYS:
            assert np.array_equal(a[k], ref[k]), (g, k)
        assert S.frequencies() == prep["freq"] and S.skipped == prep["skipped"]


def test_truncated_gzip_is_an_error_not_a_shorter_index(tmp_path):
    """a .gz cut in the middle (or with damaged data) must fail as the reference's gzip.open does, not
    yield MORNA_OK with fewer lines; with and without the reader / tokeniser threads."""
    d = synthetic_intropolis(600, J=3000)
    path = str(tmp_path / "i.tsv.gz")
    with gzip.open(path, "wt") as fh:
        for j, k in enumerate(d["keys"]):
            lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
            fh.write("\t".join(k.split(" ") + ["+", "GT", "AG", ",".join(map(str, d["samples"][lo:hi])),
                                               ",".join(map(str, d["cov"][lo:hi]))]) + "\n")
    whole = mindex.ParsedLines(path, 600, 10)
    raw = open(path, "rb").read()
    cut = str(tmp_path / "cut.tsv.gz")
    open(cut, "wb").write(raw[:len(raw) // 2])
    bad = bytearray(raw)
    for i in range(len(raw) // 2, len(raw) // 2 + 64):
        bad[i] ^= 0x5A
    dmg = str(tmp_path / "dmg.tsv.gz")
    open(dmg, "wb").write(bytes(bad))
    for threads in ("1", "4"):
        os.environ["MORNA_PARSE_THREADS"] = threads
        try:
            for p in (cut, dmg):
                for sc in (600, None):                 # with -s, and through the count_samples pass
                    with pytest.raises(IOError):
                        mindex.ParsedLines(p, sc, 10)
            again = mindex.ParsedLines(path, 600, 10)  # the undamaged file still parses, same result
            assert (again.n_lines, again.nnz, again.n_items) == (whole.n_lines, whole.nnz, whole.n_items)
        finally:
            del os.environ["MORNA_PARSE_THREADS"]
