"""`morna index` / `morna search` end to end on the GPU, read like the reference's
embedded tests (morna.py:1151-1330): go_index on the inline fixtures, then the
hard-coded neighbour orderings through the saved index.  -m gpu"""
import gzip
import io
import os

import numpy as np
import pytest

from conftest import angular64, assert_tie_aware_order

pytestmark = pytest.mark.gpu


def _write_gz(path, lines):
    with gzip.open(path, "wt") as fh:
        fh.write("".join(lines))


@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
def test_go_index_then_search_member(tmp_path, embedded, embedded_mats, name):
    from morna_amd.index import go_index
    from morna_amd.search import MornaSearch
    spec = embedded["expected"][name]
    src = str(tmp_path / "junctions.temp")
    _write_gz(src, embedded[spec["input"]])
    base = str(tmp_path / "tempIndex")
    idx = go_index(intropolis=src, basename=base, features=3000, n_trees=20, sample_count=10,
                   sample_threshold=spec["sample_threshold"], buffer_size=1024, verbose=False, metafile=None)
    assert idx.get_n_items() == spec["n_items"]
    for ext in (".annoy.mor", ".stats.mor", ".freq.mor", ".map.mor"):
        assert os.path.exists(base + ext)
    with open(base + ".stats.mor") as fh:
        assert fh.read().split() == ["10", str(spec["n_items"]), "3000"]
    X = embedded_mats["%s_D3000_f32" % name]
    s = MornaSearch(basename=base)
    assert s.annoy_index.get_n_items() == spec["n_items"]
    assert s.annoy_index.get_items().tobytes() == X.tobytes()
    inv = {v: k for k, v in s.internal_id_map.items()}
    for i, exp in enumerate(spec["orderings"]):
        res = s.search_member_n(inv[i], 10, 100, include_distances=False)      # by sample id, as morna -q
        assert isinstance(res, tuple) and len(res) == 1
        assert_tie_aware_order(res[0], exp, angular64(X, i), tol=2e-6)
        ids, d = s.search_member_n(inv[i], 10, 100, include_distances=True)
        assert np.allclose(d, np.sqrt(np.maximum(angular64(X, i)[ids], 0)), atol=2e-6)
    with pytest.raises(ValueError):
        s.search_member_n(424242, 10, 100)


def test_native_prepass_builds_the_same_index(tmp_path):
    """go_index with the library's tokenising pre-pass == go_index with the Python loop."""
    import pickle
    from conftest import GOLDEN
    from morna_amd.index import go_index
    src = str(tmp_path / "tiny.tsv.gz")
    with open(os.path.join(GOLDEN, "tiny_intropolis.tsv")) as fh:
        _write_gz(src, fh.readlines())
    a = go_index(src, str(tmp_path / "py"), 128, 4, None, 100, 1024, False, None, native=False)
    b = go_index(src, str(tmp_path / "nat"), 128, 4, None, 100, 1024, False, None, native=True)
    assert a.sample_count == b.sample_count == 6850
    assert a.get_items().tobytes() == b.get_items().tobytes()
    assert a.internal_id_map == b.internal_id_map and dict(a.sample_frequencies) == dict(b.sample_frequencies)
    items = np.arange(64, dtype=np.int32)
    ra, rb = a.get_nns_by_item_batch(items, 10, 100), b.get_nns_by_item_batch(items, 10, 100)
    assert ra[0].tolist() == rb[0].tolist() and ra[1].tobytes() == rb[1].tobytes()
    for ext in (".stats.mor", ".freq.mor", ".map.mor"):
        with open(str(tmp_path / "py") + ext, "rb") as f1, open(str(tmp_path / "nat") + ext, "rb") as f2:
            if ext == ".stats.mor":
                assert f1.read() == f2.read()
            else:
                assert pickle.load(f1) == pickle.load(f2)


def test_cli_index_and_stream_search(tmp_path, embedded):
    from morna_amd import cli
    from oracle import morna_ref
    src = str(tmp_path / "j.gz")
    _write_gz(src, embedded["generic"])
    base = str(tmp_path / "idx")
    assert cli.main(["index", "--intropolis", src, "-x", base, "--features", "128", "--n-trees", "5",
                     "-t", "1"]) == 0                                  # sample count is counted (no -s)
    # a raw-format query built from sample 3's junctions (+ one junction the index never saw)
    lines = []
    for ln in embedded["generic"]:
        key, samples, cov = morna_ref.tokenize_line(ln)
        if 3 in samples:
            c, a, b = key.split(" ")
            lines.append("%s\t%s\t%s\t%d\n" % (c, a, b, cov[samples.index(3)]))
    lines.append("chr7\t1\t2\t5\n")
    # reference-side answer: RefSearch over the same matrix
    ref_idx = morna_ref.go_index_lines(embedded["generic"], 128, None, 1)
    rs = morna_ref.RefSearch(ref_idx.sample_count, 128, ref_idx.sample_frequencies, ref_idx.matrix32())
    for ln in lines:
        t = ln.strip().split("\t")
        if " ".join(t[:3]) in ref_idx.sample_frequencies:
            rs.update_query((t[0], int(t[1]), int(t[2]), int(t[3])))
    rs.finalize_query()
    want_ids, want_d = rs.exact_search_nn(4, include_distances=True)
    out = io.StringIO()
    assert cli.main(["search", "-x", base, "-f", "raw", "--exact", "-d", "-r", "4"],
                    stdin=io.StringIO("".join(lines)), stdout=out) == 0
    got = [ln.split("\t") for ln in out.getvalue().strip().split("\n")]
    assert [g[0] for g in got] == ["1.", "2.", "3.", "4."]             # results_output format
    assert [int(g[1]) for g in got] == want_ids
    assert [float(g[2]) for g in got] == want_d                        # fp64 distances, bit-exact through repr
    assert want_ids[0] == ref_idx.internal_id_map[3]
    # approximate path prints ids only without -d
    out = io.StringIO()
    assert cli.main(["search", "-x", base, "-f", "raw", "-r", "3"], stdin=io.StringIO("".join(lines)), stdout=out) == 0
    rows = out.getvalue().strip().split("\n")
    assert len(rows) == 3 and rows[0].split("\t") == ["1.", str(want_ids[0])]
    # -q: by sample id
    out = io.StringIO()
    assert cli.main(["search", "-x", base, "-q", "3", "-r", "2", "-d"], stdout=out) == 0
    assert out.getvalue().split("\n")[0].split("\t")[1] == str(ref_idx.internal_id_map[3])


def test_metadata_indexing(tmp_path, embedded, capsys):
    """morna.py:1195-1240 test_metadata_indexing: go_index with a metafile, then
    search_member_n(1, 10, 100, include_distances=False, meta_db=True) -> (ids, fetchone() tuples)."""
    from morna_amd.index import go_index
    from morna_amd.search import MornaSearch
    from morna_amd import cli
    src, meta = str(tmp_path / "junctions.temp"), str(tmp_path / "meta.temp")
    _write_gz(src, embedded["generic"])
    with open(meta, "w") as fh:
        fh.write("".join(embedded["meta"]))
    base = str(tmp_path / "tempIndex")
    go_index(intropolis=src, basename=base, features=3000, n_trees=20, sample_count=10, sample_threshold=1,
             buffer_size=1024, verbose=False, metafile=meta)
    assert os.path.exists(base + ".meta.mor")
    exp = embedded["meta_expected"]
    s = MornaSearch(basename=base)
    results = s.search_member_n(exp["query_sample_id"], 10, 100, include_distances=False, meta_db=True)
    assert results == (exp["ids"], [tuple(k) for k in exp["keywords"]])
    ids, d, kw = s.search_member_n(exp["query_sample_id"], 10, 100, include_distances=True, meta_db=True)
    assert ids == exp["ids"] and kw == [tuple(k) for k in exp["keywords"]] and len(d) == 10
    s.query_sample = [float(x) for x in s.annoy_index.get_item_vector(0)]
    assert s.exact_search_nn(3, include_distances=False, meta_db=True)[1][0] == tuple(exp["keywords"][0])
    assert s.search_nn(3, 100, include_distances=False, meta_db=True)[1][0] == tuple(exp["keywords"][0])
    # the command line: index -m <file>, search -m
    base2 = str(tmp_path / "cliIndex")
    assert cli.main(["index", "--intropolis", src, "-x", base2, "--features", "3000", "--n-trees", "20", "-s", "10",
                     "-t", "1", "-m", meta]) == 0
    out = io.StringIO()
    assert cli.main(["search", "-x", base2, "-q", "1", "-r", "10", "-m"], stdout=out) == 0
    rows = [ln.split("\t", 2) for ln in out.getvalue().rstrip("\n").split("\n")]
    assert [int(r[1]) for r in rows] == exp["ids"]
    assert rows[0][2] == str(tuple(exp["keywords"][0]))               # results_output prints str(fetchone())


def _sam_lines(junctions):
    """One spliced alignment per (chrom, start, end) repeated cov times (utils.py:194-241 coordinates)."""
    out = []
    for chrom, start, end, cov in junctions:
        for _ in range(cov):
            out.append("r\t0\t%s\t%d\t255\t10M%dN10M\t*\t0\t0\t*\t*\n" % (chrom, start - 10, end - start + 1))
    return out


def test_cli_convergence_backoff(tmp_path, embedded):
    """-c / -ch (morna.py:1378-1452): search at junction checkpoint, checkpoint + c, + 2c, ...; stop when the
    result ids repeat; at stream end without convergence search the whole query."""
    from morna_amd import cli, streams
    from oracle import morna_ref
    src = str(tmp_path / "j.gz")
    _write_gz(src, embedded["generic"])
    base = str(tmp_path / "idx")
    assert cli.main(["index", "--intropolis", src, "-x", base, "--features", "3000", "--n-trees", "10", "-s", "10",
                     "-t", "1"]) == 0
    juncs = []
    for ln in embedded["generic"]:
        key, samples, cov = morna_ref.tokenize_line(ln)
        if 8 in samples:
            c, a, b = key.split(" ")
            juncs.append((c, int(a), int(b), cov[samples.index(8)]))
    sam = _sam_lines(juncs)
    parsed = list(streams.junctions_from_sam_stream(io.StringIO("".join(sam))))
    total = {}
    for j in parsed:
        total[tuple(j[:3])] = total.get(tuple(j[:3]), 0) + j[3]
    assert total == {(c, a, b): cov for c, a, b, cov in juncs}

    def run(extra):
        out = io.StringIO()
        assert cli.main(["search", "-x", base, "-f", "sam", "-r", "3"] + extra, stdin=io.StringIO("".join(sam)),
                        stdout=out) == 0
        return [int(ln.split("\t")[1]) for ln in out.getvalue().strip().split("\n")]
    full = run([])
    assert full[0] == 7                                                # sample 8 is internal id 7
    # expected: replay the loop through the MornaSearch methods on prefixes of the junction stream (the
    # zero-overlap samples tie at sqrt(2), so the replay must break ties as the searched index does)
    from morna_amd.search import MornaSearch

    def ref_search(prefix):
        rs = MornaSearch(basename=base)
        for j in prefix:
            rs.update_query(j)
        rs.finalize_query()
        return rs.search_nn(3, 100, include_distances=False)[0]
    assert ref_search(parsed) == full
    for c, ch in ((1, 0), (2, 1), (1000, 0), (1000, 5000)):
        backoff, checkpoint, old, want = c, ch, [-1, -1, -1], None
        for i in range(len(parsed)):
            if i == checkpoint:
                checkpoint += backoff
                backoff += backoff
                res = ref_search(parsed[:i + 1])
                if res == old:
                    want = res
                    break
                old = res
        if want is None:
            want = ref_search(parsed)
        got = run(["-c", str(c), "-ch", str(ch)])
        assert got == want, (c, ch, got, want)


def test_cli_index_from_pretokenised_cache(tmp_path, embedded):
    """`index --cache`: the second run builds from the cache alone and yields the same index files."""
    from morna_amd import cli
    from morna_amd.search import MornaSearch
    src = str(tmp_path / "j.gz")
    _write_gz(src, embedded["generic"])
    cache = str(tmp_path / "j.cache")
    common = ["index", "--intropolis", src, "--features", "128", "--n-trees", "6", "-t", "1"]
    assert cli.main(common + ["-x", str(tmp_path / "plain")]) == 0
    assert cli.main(common + ["-x", str(tmp_path / "c1"), "--cache", cache]) == 0
    stamp = os.stat(cache).st_mtime_ns
    assert cli.main(common + ["-x", str(tmp_path / "c2"), "--cache", cache]) == 0
    assert os.stat(cache).st_mtime_ns == stamp                         # reused, not rewritten
    for ext in (".annoy.mor", ".stats.mor", ".freq.mor", ".map.mor"):
        blobs = [open(str(tmp_path / b) + ext, "rb").read() for b in ("plain", "c1", "c2")]
        assert blobs[0] == blobs[1] == blobs[2], ext
    assert MornaSearch(str(tmp_path / "c2")).search_member_n(3, 4, 100)[0][0] == 2


def test_cli_index_shards_then_search_equals_unsharded(tmp_path):
    """`morna index --shards 3`: one file cut into three row shards with the GLOBAL idf and first-seen
    ids, each with its own matrix + forest file; `morna search` on that file set answers as on the unsharded index --
    the exact search identically (ids and fp64 distances, stream query and --exact), by-member and approximate
    queries with the same nearest neighbour and global ids."""
    import pickle
    from morna_amd import cli
    from morna_amd.index import shard_bounds, shard_basename
    from morna_amd.search import MornaSearch
    from morna_amd.synth import synthetic_intropolis
    d = synthetic_intropolis(700, J=900)
    src = str(tmp_path / "i.tsv.gz")
    lines = []
    for j, k in enumerate(d["keys"]):
        lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
        lines.append("\t".join(k.split(" ") + ["+", "GT", "AG", ",".join(map(str, d["samples"][lo:hi])),
                                               ",".join(map(str, d["cov"][lo:hi]))]) + "\n")
    _write_gz(src, lines)
    one, three = str(tmp_path / "one"), str(tmp_path / "three")
    assert cli.main(["index", "--intropolis", src, "-x", one, "--features", "96", "--n-trees", "6", "-t", "40"]) == 0
    assert cli.main(["index", "--intropolis", src, "-x", three, "--features", "96", "--n-trees", "6", "-t", "40",
                     "--shards", "3"]) == 0
    for ext in (".stats.mor", ".freq.mor", ".map.mor"):
        with open(one + ext, "rb") as f1, open(three + ext, "rb") as f2:
            assert (f1.read() == f2.read()) if ext == ".stats.mor" else (pickle.load(f1) == pickle.load(f2))
    s1, s3 = MornaSearch(basename=one), MornaSearch(basename=three)
    N = s1.index_size
    assert s3.annoy_index.get_n_items() == N and s3.annoy_index.offsets.tolist() == shard_bounds(N, 3)
    X = s1.annoy_index.get_items()
    stacked = np.concatenate([sh.get_items() for sh in s3.annoy_index.shards])
    assert stacked.tobytes() == X.tobytes()
    for g in range(3):
        assert os.path.exists(shard_basename(three, g, 3) + ".annoy.mor")
    # a stream query made of one sample's junctions
    sample = int(d["samples"][d["row_ptr"][5]])
    q = []
    for j, k in enumerate(d["keys"]):
        lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
        hit = np.nonzero(d["samples"][lo:hi] == sample)[0]
        if len(hit):
            c, a, b = k.split(" ")
            q.append("%s\t%s\t%s\t%d\n" % (c, a, b, d["cov"][lo + hit[0]]))
    outs = []
    for base in (one, three):
        out = io.StringIO()
        assert cli.main(["search", "-x", base, "-f", "raw", "--exact", "-d", "-r", "8"], stdin=io.StringIO("".join(q)), stdout=out) == 0
        outs.append(out.getvalue())
    assert outs[0] == outs[1] and len(outs[0].strip().split("\n")) == 8
    a1 = s1.search_member_n(sample, 10, -1, include_distances=True)
    a3 = s3.search_member_n(sample, 10, -1, include_distances=True)
    assert a1[0][0] == a3[0][0] == s1.internal_id_map[sample] and a3[1][0] < 1e-3
    # search_k = -1 with 6 trees per forest inspects 60 candidates per forest: the sharded answer's distances are the true
    # distances of its (global) ids
    assert np.allclose(a3[1], np.sqrt(np.maximum(angular64(X, a3[0][0])[a3[0]], 0)), atol=2e-6)
    assert s3.annoy_index.get_item_vector(N - 1) == s1.annoy_index.get_item_vector(N - 1)
    with pytest.raises(IndexError):
        s3.annoy_index.get_item_vector(N)


def test_cli_index_shards_one_process_per_rank_under_torchrun(tmp_path):
    """`morna index --shards 2` launched as `torch.distributed.run --nproc-per-node 2`: every rank parses the file (both
    write the pre-tokenised cache: it is renamed into place, so neither reads half of the other's), builds ITS shard on its
    device (here both share the one GPU) and saves it; rank 0 writes the global files.  The file set equals the one a single
    process writes, and its stacked matrices are the unsharded index's."""
    import socket
    import subprocess
    import sys
    from morna_amd import cli
    from morna_amd.search import MornaSearch
    from morna_amd.synth import synthetic_intropolis
    d = synthetic_intropolis(900, J=1200)
    src = str(tmp_path / "i.tsv.gz")
    lines = []
    for j, k in enumerate(d["keys"]):
        lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
        lines.append("\t".join(k.split(" ") + ["+", "GT", "AG", ",".join(map(str, d["samples"][lo:hi])),
                                               ",".join(map(str, d["cov"][lo:hi]))]) + "\n")
    _write_gz(src, lines)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    two, one = str(tmp_path / "two"), str(tmp_path / "one")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "morna_amd.cli", "index", "--intropolis", src, "-x", two,
                        "--features", "96", "--n-trees", "5", "-t", "40", "--shards", "2", "--cache", str(tmp_path / "i.cache")],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert cli.main(["index", "--intropolis", src, "-x", one, "--features", "96", "--n-trees", "5", "-t", "40"]) == 0
    a, b = MornaSearch(one), MornaSearch(two)
    assert b.annoy_index.offsets.tolist() == [0, 450, 900]
    assert np.concatenate([sh.get_items() for sh in b.annoy_index.shards]).tobytes() == a.annoy_index.get_items().tobytes()
    assert a.internal_id_map == b.internal_id_map and dict(a.sample_frequencies) == dict(b.sample_frequencies)
    q = a.annoy_index.get_item_vector(123)
    a.query_sample = b.query_sample = [float(v) for v in q]
    assert a.exact_search_nn(8) == b.exact_search_nn(8)


def test_cli_search_one_process_per_shard_under_torchrun(tmp_path):
    """`morna search` launched as `torch.distributed.run --nproc-per-node 2` on a file set of two shards: every rank loads ITS
    shard, the ranks answer together (dist.ShardedSearch: RCCL inside the library when every rank has a GPU, gloo through
    the host when they share one, as here) and rank 0 prints -- the lines the single process prints over the same file
    set, by member and for a raw-format stream with --exact."""
    import socket
    import subprocess
    import sys
    from morna_amd import cli
    from morna_amd.synth import synthetic_intropolis
    d = synthetic_intropolis(900, J=1200)
    src = str(tmp_path / "i.tsv.gz")
    lines = []
    for j, k in enumerate(d["keys"]):
        lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
        lines.append("\t".join(k.split(" ") + ["+", "GT", "AG", ",".join(map(str, d["samples"][lo:hi])),
                                               ",".join(map(str, d["cov"][lo:hi]))]) + "\n")
    _write_gz(src, lines)
    base = str(tmp_path / "two")
    assert cli.main(["index", "--intropolis", src, "-x", base, "--features", "96", "--n-trees", "5", "-t", "40", "--shards", "2"]) == 0
    sample = int(d["samples"][d["row_ptr"][7] + 3])
    q = []
    for j, k in enumerate(d["keys"]):
        lo, hi = d["row_ptr"][j], d["row_ptr"][j + 1]
        hit = np.nonzero(d["samples"][lo:hi] == sample)[0]
        if len(hit):
            c, a, b = k.split(" ")
            q.append("%s\t%s\t%s\t%d\n" % (c, a, b, d["cov"][lo + hit[0]]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def torchrun(argv, stdin_text=None):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), "-m", "morna_amd.cli"] + argv, cwd=root, input=stdin_text,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [ln for ln in r.stdout.splitlines() if ln[:1].isdigit()]

    def single(argv, stdin_text=None):
        out = io.StringIO()
        assert cli.main(argv, stdin=io.StringIO(stdin_text or ""), stdout=out) == 0
        return [ln for ln in out.getvalue().splitlines() if ln[:1].isdigit()]

    by_member = ["search", "-x", base, "-q", str(sample), "-d", "-r", "8", "--search-k", "-1"]
    assert torchrun(by_member) == single(by_member) and len(single(by_member)) == 8
    stream = ["search", "-x", base, "-f", "raw", "--exact", "-d", "-r", "8"]
    assert torchrun(stream, "".join(q)) == single(stream, "".join(q))
