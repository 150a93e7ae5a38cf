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
