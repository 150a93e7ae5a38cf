#!/usr/bin/env python3
"""End-to-end `morna index` on a synthetic intropolis FILE of BASELINE.json configs[2] size (50k samples x 70k junction
lines, ~1e8 (sample, coverage) pairs, ~0.8 GB of text): what go_index (morna.py:824-865) costs from the file on disk to
the saved index, stage by stage.  Never part of bench.py's `value` (that is measured with the lines resident in HBM).

    python scripts/bench_e2e_index.py [--samples 50000] [--junctions 70000] [--trees 200] [--out profiles/r03_e2e_index.json]

Stages timed:
  write     (setup, not part of indexing) the synthetic lines as text, plain and gzipped
  parse     morna_parse_intropolis: the native host pre-pass (tokenise, threshold, first-seen ids, idf)  -- MB/s of text
  stage     host -> HBM copy of the parsed arrays (+ the item order); pageable host memory, and the two big arrays pinned
  features  feature build on the GPU
  forest    build(n_trees)
  save      the index file set
and the same through `python -m morna_amd.cli index` as one wall-clock number (a fresh process: imports, parse, build, save).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50_000)
    ap.add_argument("--junctions", type=int, default=70_000)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--features", type=int, default=3000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from morna_amd import index as mindex
    from morna_amd._lib import check, lib, ptr
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.synth import synthetic_intropolis

    res = {"config": {"samples": a.samples, "junction_lines": a.junctions, "features": a.features, "trees": a.trees}}
    data = synthetic_intropolis(a.samples, J=a.junctions)
    keys = [k.encode("ascii") for k in data["keys"]]
    key_off = np.zeros(len(keys) + 1, np.int64)
    key_off[1:] = np.cumsum([len(k) for k in keys])
    key_bytes = np.frombuffer(b"".join(keys), np.uint8)
    samples = np.ascontiguousarray(data["samples"], np.int64)
    cov = np.ascontiguousarray(data["cov"], np.int32)
    row_ptr = np.ascontiguousarray(data["row_ptr"], np.int64)
    res["config"]["nnz"] = int(len(samples))
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        files = {}
        for name in ("plain.tsv", "gz.tsv.gz"):
            path = os.path.join(tmp, name)
            t0 = time.perf_counter()
            check(lib().morna_write_intropolis(path.encode(), ptr(key_bytes), ptr(key_off), len(keys), ptr(row_ptr), ptr(samples),
                                               ptr(cov)))
            files[name] = (path, time.perf_counter() - t0, os.path.getsize(path))
        text_bytes = files["plain.tsv"][2]
        res["file"] = {"text_bytes": text_bytes, "gz_bytes": files["gz.tsv.gz"][2],
                       "write_s": {k: v[1] for k, v in files.items()}}
        # ---- stage by stage, in this process
        for name in ("gz.tsv.gz", "plain.tsv"):
            path = files[name][0]
            t0 = time.perf_counter()
            parsed = mindex.ParsedLines(path, a.samples, 100)
            t_parse = time.perf_counter() - t0
            st = {"parse_s": t_parse, "text_MB_per_s": text_bytes / 1e6 / t_parse, "lines_per_s": parsed.lines_read / t_parse,
                  "pairs_per_s": parsed.nnz / t_parse}
            if name == "plain.tsv":
                res["parse_plain"] = st
                continue
            res["parse_gz"] = st
            idx = AnnoyIndex(a.features)
            t0 = time.perf_counter()
            parsed.stage(idx)
            idx.synchronize()
            t_stage = time.perf_counter() - t0
            arrs = parsed.arrays()
            staged_bytes = sum(arrs[k].nbytes for k in ("key_bytes", "key_off", "row_ptr", "ids", "cov", "idf"))
            # the same copy with the two big arrays page-locked first (hipHostRegister): what a caller that keeps its
            # parse buffers pinned would see; the registration itself is timed apart
            hip = C.CDLL("libamdhip64.so")
            hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
            hip.hipHostUnregister.argtypes = [C.c_void_p]
            t0 = time.perf_counter()
            pinned = [hip.hipHostRegister(arrs[k].ctypes.data, arrs[k].nbytes, 0) for k in ("ids", "cov")]
            t_pin = time.perf_counter() - t0
            t0 = time.perf_counter()
            parsed.stage(idx)
            idx.synchronize()
            t_stage_pinned = time.perf_counter() - t0
            for k in ("ids", "cov"):
                hip.hipHostUnregister(arrs[k].ctypes.data)
            t0 = time.perf_counter()
            idx.build_features(parsed.n_items)
            idx.synchronize()
            t_feat = time.perf_counter() - t0
            idx.unstage_junctions()
            t0 = time.perf_counter()
            idx.build(a.trees)
            idx.synchronize()
            t_forest = time.perf_counter() - t0
            t0 = time.perf_counter()
            idx.save(os.path.join(tmp, "x.annoy.mor"))
            t_save = time.perf_counter() - t0
            res["stages_s"] = {"parse_gz": t_parse, "stage_pageable": t_stage, "stage_pinned": t_stage_pinned,
                               "pin_registration": t_pin, "features": t_feat, "forest": t_forest, "save_annoy_blob": t_save}
            res["stage_GB_per_s"] = {"pageable": staged_bytes / 1e9 / t_stage, "pinned": staged_bytes / 1e9 / t_stage_pinned,
                                     "bytes": staged_bytes, "pin_rc": [int(p) if isinstance(p, int) else str(p) for p in pinned]}
            del idx
        # ---- the CLI, one fresh process
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "morna_amd.cli", "index", "--intropolis", files["gz.tsv.gz"][0], "-x",
                            os.path.join(tmp, "cli"), "-s", str(a.samples), "--n-trees", str(a.trees), "--features",
                            str(a.features)], cwd=ROOT, capture_output=True, text=True)
        res["cli_index_wall_s"] = time.perf_counter() - t0
        res["cli_rc"] = r.returncode
        if r.returncode != 0:
            res["cli_stderr"] = r.stderr[-500:]
        res["samples_per_s_end_to_end_cli"] = a.samples / res["cli_index_wall_s"]
        # ---- `morna search` end to end on that index: one query per process, as the reference is used (morna.py:1345-1484)
        if r.returncode == 0:
            from morna_amd.search import MornaSearch
            t0 = time.perf_counter()
            ms = MornaSearch(basename=os.path.join(tmp, "cli"))
            t_load = time.perf_counter() - t0
            some_id = next(iter(ms.internal_id_map))
            t0 = time.perf_counter()
            ms.annoy_index.get_nns_by_item(ms.internal_id_map[some_id], 20, 100, include_distances=True)
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            ms.annoy_index.get_nns_by_item(ms.internal_id_map[some_id], 20, 100, include_distances=True)
            t_second = time.perf_counter() - t0
            del ms
            t0 = time.perf_counter()
            r2 = subprocess.run([sys.executable, "-m", "morna_amd.cli", "search", "-x", os.path.join(tmp, "cli"), "-q", str(some_id),
                                 "-d"], cwd=ROOT, capture_output=True, text=True)
            res["cli_search_by_member"] = {"wall_s": time.perf_counter() - t0, "rc": r2.returncode,
                                           "result_lines": len([ln for ln in r2.stdout.splitlines() if ln[:1].isdigit()]),
                                           "in_process": {"load_index_file_set_s": t_load, "first_query_s": t_first,
                                                          "second_query_s": t_second},
                                           "index_bytes": os.path.getsize(os.path.join(tmp, "cli.annoy.mor"))}
        # ---- what the binary pre-tokenised cache buys: `index --cache` twice over the same file -- the first
        # run parses and writes the cache, the second (other --n-trees, as a user tuning the index would) reads it back
        cache = os.path.join(tmp, "lines.cache")
        runs = {}
        for name, trees in (("first_run_writes_cache", a.trees), ("second_run_reads_cache", max(a.trees // 2, 1))):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "morna_amd.cli", "index", "--intropolis", files["gz.tsv.gz"][0], "-x",
                                os.path.join(tmp, "cli_" + name), "-s", str(a.samples), "--n-trees", str(trees), "--features",
                                str(a.features), "--cache", cache], cwd=ROOT, capture_output=True, text=True)
            runs[name] = {"wall_s": time.perf_counter() - t0, "rc": r.returncode, "n_trees": trees}
        runs["cache_bytes"] = os.path.getsize(cache) if os.path.exists(cache) else None
        t0 = time.perf_counter()
        again = mindex.ParsedLines(files["gz.tsv.gz"][0], a.samples, 100, cache=cache)
        runs["cache_load_s_in_process"] = time.perf_counter() - t0
        runs["loaded_from_cache"] = bool(again.from_cache)
        res["cli_index_with_cache"] = runs
        # ---- the same file as 8 row shards built one after the other by one process (`index --shards 8`)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "morna_amd.cli", "index", "--intropolis", files["gz.tsv.gz"][0], "-x",
                            os.path.join(tmp, "cli_shards"), "-s", str(a.samples), "--n-trees", str(a.trees), "--features",
                            str(a.features), "--cache", cache, "--shards", "8"], cwd=ROOT, capture_output=True, text=True)
        res["cli_index_8_shards_from_cache"] = {"wall_s": time.perf_counter() - t0, "rc": r.returncode}
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
