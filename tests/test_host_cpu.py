"""Host-side mirror of the reference interface: parsers, tokenising, the
vectorised add_junction pre-pass.  No GPU needed."""
import io

import numpy as np
import pytest

from morna_amd import index as mindex
from morna_amd import streams
from oracle import morna_ref


def test_count_samples_and_tokenize(embedded):
    assert mindex.count_samples(embedded["generic"]) == 10          # morna.py:1140-1149
    k, s, c = mindex.tokenize_line(embedded["generic"][16])
    assert k == "chr10 102161920 102162275"
    assert s == list(range(1, 11)) and c == [2, 1, 1, 2, 2, 1, 2, 3, 8, 2]
    assert mindex.tokenize_line(embedded["generic"][16]) == morna_ref.tokenize_line(embedded["generic"][16])


def _tok(lines):
    keys, rp, s, c = [], [0], [], []
    for ln in lines:
        k, ss, cc = mindex.tokenize_line(ln)
        keys.append(k)
        s += ss
        c += cc
        rp.append(len(s))
    return keys, np.array(rp, np.int64), np.array(s, np.int64), np.array(c, np.int64)


@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
def test_prepare_csr_equals_reference_prepass(embedded, name):
    """threshold, cumulative freq, first-seen ids and idf as MornaIndex.add_junction computes them"""
    spec = embedded["expected"][name]
    lines = embedded[spec["input"]]
    ref = morna_ref.go_index_lines(lines, 128, spec["sample_count"], spec["sample_threshold"])
    keys, rp, s, c = _tok(lines)
    prep = mindex.prepare_csr(keys, rp, s, c, spec["sample_count"], spec["sample_threshold"])
    assert prep["n_items"] == ref.new_internal_id == spec["n_items"]
    assert prep["skipped"] == ref.skipped
    assert prep["ext_ids"].tolist() == sorted(ref.internal_id_map, key=ref.internal_id_map.get)
    assert prep["freq"] == dict(ref.sample_frequencies)
    assert len(prep["idf"]) == len(lines) - ref.skipped
    assert prep["row_ptr"][-1] == len(prep["ids"]) == len(prep["cov"])
    assert prep["ids"].max() == prep["n_items"] - 1


def test_prepare_csr_duplicate_keys_use_cumulative_frequency():
    from math import log
    keys = ["chr1 1 2", "chr1 5 9", "chr1 1 2"]
    rp = np.array([0, 2, 4, 7])
    s = np.array([7, 9, 9, 3, 7, 3, 11])
    c = np.ones(7, np.int64)
    prep = mindex.prepare_csr(keys, rp, s, c, 20, 1)
    assert prep["idf"].tolist() == [log(20.0 / 2), log(20.0 / 2), log(20.0 / 5)]      # morna.py:365, 372
    assert prep["ext_ids"].tolist() == [7, 9, 3, 11]
    assert prep["ids"].tolist() == [0, 1, 1, 2, 0, 2, 3]


def test_raw_stream():
    got = list(streams.junctions_from_raw_stream(io.StringIO("chr1\t100\t200\t7\nchrX\t5\t9\t1\textra\n")))
    assert got == [("chr1", 100, 200, 7), ("chrX", 5, 9, 1)]


def test_bed_stream():
    # TopHat-style junctions.bed: blocks (size 10 at 0) and (size 20 at 110) of a feature starting at 1000
    line = "chr2\t1000\t1130\tJUNC1\t42\t+\t1000\t1130\t255,0,0\t2\t10,20\t0,110\n"
    assert list(streams.junctions_from_bed_stream(io.StringIO(line))) == [("chr2", 1011, 1110, 42)]
    # three blocks -> two junctions; trailing commas; a short line is skipped
    line3 = "chr2\t1000\t1400\tJ\t3\t-\t1000\t1400\t0\t3\t10,20,30,\t0,100,370,\nshort\tline\n"
    assert list(streams.junctions_from_bed_stream(io.StringIO(line3))) == [("chr2", 1011, 1100, 3),
                                                                           ("chr2", 1121, 1370, 3)]


def test_sam_stream():
    sam = ("@HD\tVN:1.0\n"
           "r1\t0\tchr1\t100\t255\t10M50N10M\t*\t0\t0\tACGTACGTACGTACGTACGT\t*\n"        # junction 110..159
           "r2\t4\tchr1\t100\t255\t10M50N10M\t*\t0\t0\tACGT\t*\n"                         # unmapped
           "r3\t256\tchr1\t100\t255\t10M50N10M\t*\t0\t0\tACGT\t*\n"                       # secondary
           "r4\t16\tchr3\t7\t255\t2S5M2I3M100N4M2D6M30N1M\t*\t0\t0\tACGTACGTACGTACGTACGTACG\t*\n"
           "r5\t0\tchr1\t100\t255\t20M\t*\t0\t0\tACGT\t*\n")                              # unspliced
    got = list(streams.junctions_from_sam_stream(io.StringIO(sam)))
    # r4: pos 7; 5M -> 12; 3M -> 15; 100N -> junction (15, 114), pos 115; 4M -> 119; 2D -> 121; 6M -> 127; 30N -> (127, 156)
    assert got == [("chr1", 110, 159, 1), ("chr3", 15, 114, 1), ("chr3", 127, 156, 1)]
    with pytest.raises(RuntimeError):
        list(streams.junctions_from_sam_stream(io.StringIO("r\t0\tc\t1\t255\t5M3X2N5M\t*\t0\t0\tAC\t*\n")))
    with pytest.raises(IndexError):
        list(streams.junctions_from_sam_stream(io.StringIO("r\t0\tc\t1\t255\t5M2N5M\n")))


def test_cli_parser_defaults_match_reference():
    from morna_amd.cli import build_parser
    p = build_parser()
    a = p.parse_args(["index", "--intropolis", "x.gz"])
    assert (a.basename, a.features, a.n_trees, a.sample_count, a.sample_threshold, a.buffer_size) == \
        ("morna", 3000, 200, None, 100, 1024)                         # morna.py:970-1021
    s = p.parse_args(["search", "-x", "idx"])
    assert (s.search_k, s.format, s.distances, s.query_id, s.exact, s.results) == (100, "sam", False, None, False, 20)
    assert (s.metadata, s.convergence_backoff, s.checkpoint, a.metafile) == (False, None, 0, None)   # morna.py:907-924, 1016


def test_metadata_db_roundtrip(tmp_path, embedded):
    """<basename>.meta.mor: the table MornaIndex.save writes and the fetchone() tuples the
    searches append (morna.py:494-520, 666-676); sample ids in the order test_metadata_indexing expects."""
    from morna_amd.metadb import lookup_meta, write_meta_db
    meta = str(tmp_path / "meta.tsv")
    with open(meta, "w") as fh:
        fh.write("".join(embedded["meta"]))
    base = str(tmp_path / "idx")
    write_meta_db(meta, base)
    write_meta_db(meta, base)                                         # overwriting an old index drops the table first
    exp = embedded["meta_expected"]
    sample_ids = [i + 1 for i in exp["ids"]]                          # "simple": internal id i <-> sample id i + 1
    got = lookup_meta(base, sample_ids)
    assert got == [tuple(k) for k in exp["keywords"]]
    assert lookup_meta(base, [4242, None]) == [None, None]            # unknown sample: fetchone() -> None
    with open(meta, "w") as fh:
        fh.write("77\n")
    with pytest.raises(IndexError):                                   # id with no keywords, as line.split(None,1)[1]
        write_meta_db(meta, str(tmp_path / "bad"))


# ---- native pre-pass (morna_parse_intropolis): no GPU involved -------------------------------

def _write(path, lines, gz):
    import gzip
    if gz:
        with gzip.open(path, "wt") as fh:
            fh.write("".join(lines))
    else:
        with open(path, "w") as fh:
            fh.write("".join(lines))


@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("gz", [True, False])
def test_native_parser_equals_python_prepass(tmp_path, embedded, name, gz):
    spec = embedded["expected"][name]
    lines = embedded[spec["input"]]
    path = str(tmp_path / ("j.tsv.gz" if gz else "j.tsv"))
    _write(path, lines, gz)
    keys, rp, s, c = _tok(lines)
    want = mindex.prepare_csr(keys, rp, s, c, spec["sample_count"], spec["sample_threshold"])
    for sc in (spec["sample_count"], None):                     # given with -s, or counted (two passes)
        got = mindex.ParsedLines(path, sc, spec["sample_threshold"])
        assert got.sample_count == 10 and got.n_items == spec["n_items"] and got.skipped == want["skipped"]
        assert got.lines_read == len(lines)
        a = got.arrays()
        for k in ("key_bytes", "key_off", "row_ptr", "ids", "cov", "ext_ids"):
            assert a[k].tolist() == np.asarray(want[k]).tolist(), k
        assert a["idf"].tobytes() == want["idf"].tobytes()       # libm log on both sides: bit-identical
        assert got.frequencies() == want["freq"]


def test_native_parser_tiny_intropolis_and_errors(tmp_path):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "tiny_intropolis_D128.npz"))
    got = mindex.ParsedLines(os.path.join(GOLDEN, "tiny_intropolis.tsv"), None, 100)
    assert got.sample_count == 6850 == int(g["sample_count"])   # counted: distinct sample-id strings
    assert got.arrays()["ext_ids"].tolist() == g["ext_ids"].tolist()
    assert got.n_lines == 3 and got.nnz == 2040 + 6210 + 1664
    # duplicate keys accumulate frequency; zip() truncation; whitespace stripping
    p = str(tmp_path / "d.tsv")
    _write(p, ["chr1\t1\t2\t+\tGT\tAG\t7,9\t1,1\n", "  chr1\t1\t2\t-\tGT\tAG\t9,3,7\t5,6\n", "chr2\t5\t6\tx\t4\t1\n"], False)
    got = mindex.ParsedLines(p, 20, 1)
    a = got.arrays()
    from math import log
    assert a["idf"].tolist() == [log(20.0 / 2), log(20.0 / 5), log(20.0 / 1)]
    assert a["ids"].tolist() == [0, 1, 1, 2, 3] and a["cov"].tolist() == [1, 1, 5, 6, 1]
    assert got.frequencies() == {"chr1 1 2": 5, "chr2 5 6": 1}
    with pytest.raises(IOError):
        mindex.ParsedLines(str(tmp_path / "missing.gz"), 5, 1)
    _write(p, ["chr1\t1\t2\t+\tGT\tAG\t7,x\t1,1\n"], False)
    with pytest.raises(ValueError):
        mindex.ParsedLines(p, 5, 1)
    _write(p, ["chr1\t1\t2\t+\tGT\tAG\t7,8\t1,4294967297\n"], False)      # a coverage the int32 arrays cannot hold
    with pytest.raises(ValueError):
        mindex.ParsedLines(p, 5, 1)


def test_add_junction_ragged_lists_follow_zip():
    """morna.py:376: `for sample_id, coverage in zip(samples, coverages)` -- the longer list is cut, the threshold
    and the frequency still use len(samples), and only the samples the loop reaches get internal ids.  The Python
    path and the native parser (test above) agree on that."""
    from math import log
    m = mindex.MornaIndex.__new__(mindex.MornaIndex)      # no device needed for the host half
    m.sample_count, m.sample_threshold, m.skipped, m.junc_id = 20, 2, 0, -1
    m.internal_id_map, m.new_internal_id = {}, 0
    from collections import defaultdict
    m.sample_frequencies = defaultdict(int)
    m._lines = mindex.JunctionBuffer()
    m.add_junction("chr1 1 2", [7, 9], [1, 1])
    m.add_junction("chr1 1 2", [9, 3, 7], [5, 6])          # coverages shorter: sample 7 is not reached here
    m.add_junction("chr2 5 6", [4, 11], [1, 2, 3])         # coverages longer: the extra one is dropped
    m.add_junction("chr3 1 9", [5], [1])                   # below the threshold
    key_bytes, key_off, row_ptr, ids, cov, idf = m._lines.arrays()
    assert row_ptr.tolist() == [0, 2, 4, 6] and ids.tolist() == [0, 1, 1, 2, 3, 4] and cov.tolist() == [1, 1, 5, 6, 1, 2]
    assert idf.tolist() == [log(20.0 / 2), log(20.0 / 5), log(20.0 / 2)]
    assert m.internal_id_map == {7: 0, 9: 1, 3: 2, 4: 3, 11: 4} and m.skipped == 1 and m.junc_id == 3
    with pytest.raises(OverflowError):
        m.add_junction("chr4 1 2", [1, 2], [1, 2 ** 31])


def test_pretokenised_cache_roundtrip_and_invalidation(tmp_path, embedded):
    """the binary cache holds exactly what the parse produced; it is reused only for the
    same file (size, mtime), sample_count argument and threshold; damaged files are refused."""
    import os
    src = str(tmp_path / "j.tsv.gz")
    _write(src, embedded["lossy"], True)
    cache = str(tmp_path / "j.cache")
    first = mindex.ParsedLines(src, 10, 4, cache=cache)
    assert not first.from_cache and os.path.exists(cache)
    again = mindex.ParsedLines(src, 10, 4, cache=cache)
    assert again.from_cache
    plain = mindex.ParsedLines(src, 10, 4)
    for k in ("n_lines", "nnz", "n_items", "skipped", "sample_count", "key_bytes_n", "n_keys", "lines_read"):
        assert getattr(again, k) == getattr(plain, k), k
    a, b = again.arrays(), plain.arrays()
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
    assert again.frequencies() == plain.frequencies()
    # other threshold / sample_count argument / touched source: parsed again, cache rewritten
    assert not mindex.ParsedLines(src, 10, 6, cache=cache).from_cache
    assert mindex.ParsedLines(src, 10, 6, cache=cache).from_cache
    assert not mindex.ParsedLines(src, None, 6, cache=cache).from_cache
    st = os.stat(src)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns + 10**9))
    assert not mindex.ParsedLines(src, None, 6, cache=cache).from_cache
    assert mindex.ParsedLines(src, None, 6, cache=cache).from_cache
    # truncated or foreign files are not trusted: parse again and overwrite
    blob = open(cache, "rb").read()
    for bad in (blob[:len(blob) // 2], b"not a cache", blob + b"x"):
        with open(cache, "wb") as fh:
            fh.write(bad)
        got = mindex.ParsedLines(src, None, 6, cache=cache)
        assert not got.from_cache and got.n_items == 7
        assert open(cache, "rb").read() == blob


def test_division_by_count_route_quick(tmp_path):
    """The fp32 FMA route two_means uses for t / (n + 1) (devutil.hpp centroid_step4) against the division, on the
    host: every float significand of t at three exponents and all subnormals (one in seven), seven divisors.
    `scripts/check_intdiv_route.c` without an argument runs all 201 divisors and eleven exponents (minutes)."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_intdiv")
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(root, "scripts", "check_intdiv_route.c"),
                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ" in r.stdout.splitlines()[0]


@pytest.mark.parametrize("gz", [False, True])
def test_write_intropolis_round_trip(tmp_path, gz):
    """morna_write_intropolis (bench / test utility) writes what morna_parse_intropolis reads back: same kept lines,
    first-seen ids, coverages and idf as the Python pre-pass over the same arrays."""
    from morna_amd._lib import check, lib, ptr
    rng = np.random.default_rng(12)
    J, n_samples = 200, 500
    keys = ["chr%d %d %d" % (rng.integers(1, 23), rng.integers(1, 10**8), rng.integers(1, 10**8)) for _ in range(J)]
    keys[17] = keys[3]
    rp, s, c = [0], [], []
    for j in range(J):
        ids = np.sort(rng.choice(n_samples, size=int(rng.integers(1, 60)), replace=False)) + 1
        s += ids.tolist()
        c += rng.integers(1, 5000, size=len(ids)).tolist()
        rp.append(len(s))
    rp, s, c = np.array(rp, np.int64), np.array(s, np.int64), np.array(c, np.int32)
    kb = [k.encode("ascii") for k in keys]
    key_off = np.zeros(J + 1, np.int64)
    key_off[1:] = np.cumsum([len(k) for k in kb])
    key_bytes = np.frombuffer(b"".join(kb), np.uint8)
    path = str(tmp_path / ("w.tsv.gz" if gz else "w.tsv"))
    check(lib().morna_write_intropolis(path.encode(), ptr(key_bytes), ptr(key_off), J, ptr(rp), ptr(s), ptr(c)))
    import gzip
    with (gzip.open(path, "rt") if gz else open(path)) as fh:
        first = fh.readline().rstrip("\n").split("\t")
    assert " ".join(first[:3]) == keys[0] and len(first) == 8
    assert first[-2] == ",".join(str(x) for x in s[:rp[1]]) and first[-1] == ",".join(str(x) for x in c[:rp[1]])
    got = mindex.ParsedLines(path, n_samples, 10)
    want = mindex.prepare_csr(keys, rp, s, c, n_samples, 10)
    a = got.arrays()
    for k in ("key_bytes", "key_off", "row_ptr", "ids", "cov", "ext_ids"):
        assert a[k].tolist() == np.asarray(want[k]).tolist(), k
    assert a["idf"].tobytes() == want["idf"].tobytes()


@pytest.mark.parametrize("gz", [False, True])
def test_native_parser_pipeline_equals_single_thread(tmp_path, monkeypatch, gz):
    """The reader / tokeniser / ordered-merge pipeline (MORNA_PARSE_THREADS > 1) gives exactly the arrays of the
    one-thread pass: blocks of 4 MB cut at line ends, lines far longer than a block, a last line without a terminator,
    duplicate keys (cumulative frequency), skipped lines, zip() truncation, counted sample_count -- and reports the
    FIRST bad line of the file, as the sequential pass does."""
    rng = np.random.default_rng(21)
    lines = []
    for j in range(900):
        n = 650_000 if j in (5, 401, 899) else int(rng.integers(1, 300))   # three lines of ~4.5 MB: longer than a block
        ids = np.sort(rng.choice(1_000_000, size=n, replace=False))
        cov = rng.integers(1, 300, size=n - (1 if j % 50 == 7 else 0))    # some lines: one coverage short
        key = "chr%d\t%d\t%d" % (j % 22 + 1, 1000 + (j % 300), 5000 + (j % 300))   # keys repeat
        lines.append("%s\t+\tGT\tAG\t%s\t%s\n" % (key, ",".join(map(str, ids.tolist())), ",".join(map(str, cov.tolist()))))
    lines[-1] = lines[-1].rstrip("\n")
    p = str(tmp_path / ("p.tsv.gz" if gz else "p.tsv"))
    _write(p, lines, gz)

    def run(threads, sample_count):
        monkeypatch.setenv("MORNA_PARSE_THREADS", str(threads))
        got = mindex.ParsedLines(p, sample_count, 50)
        a = got.arrays()
        return ([got.n_lines, got.nnz, got.n_items, got.skipped, got.sample_count, got.lines_read],
                {k: a[k].tobytes() for k in a}, got.frequencies())
    one = run(1, None)
    assert one[0][5] == 900 and one[0][3] > 0
    for threads in (2, 5):
        many = run(threads, None)
        assert many[0] == one[0] and many[2] == one[2]
        for k in one[1]:
            assert many[1][k] == one[1][k], k
    # the first bad line wins, whichever block a worker finishes first
    bad = list(lines)
    bad[-1] += "\n"
    bad[400] = bad[400].replace(",", ",x", 1)     # (line 401 of the file)
    bad[700] = "justonecolumn\n"
    pb = str(tmp_path / ("b.tsv.gz" if gz else "b.tsv"))
    _write(pb, bad, gz)
    for threads in (1, 4):
        monkeypatch.setenv("MORNA_PARSE_THREADS", str(threads))
        with pytest.raises(ValueError, match="line 401 "):
            mindex.ParsedLines(pb, 5, 1)
