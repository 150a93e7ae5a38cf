"""bench.py's own contract under -m gpu: one JSON line, the strong-scaling cut (ONE data set over N ranks)
rehearsed with two ranks on one device over gloo, and that the merged answers of the sharded run ARE the single run's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--samples", "6000", "--junctions", "5000", "--trees", "12", "--queries", "96", "--steps", "2", "--warmup", "1",
         "--no-cpu-baseline", "--no-extras", "--verify"]


def _line(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]               # ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_strong_scaling_two_rank_rehearsal_equals_the_single_run():
    one = _line([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong"
    assert one["config"]["samples_total"] == one["config"]["samples_per_gpu"] == 6000
    for key in ("roofline", "rooflines", "cpu_baseline", "verify", "ms_per_step", "value"):
        assert key in one
    assert set(one["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert one["roofline"]["frac"] > 0 and (one["roofline"]["frac"] <= 1 or "served_from_cache" in one["roofline"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo"] + SMALL)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["samples_total"] == 6000 and two["config"]["samples_per_gpu"] == 3000      # ONE data set, cut in two
    assert "cut into 2 row shards" in two["config"]["workload"]
    # the sharded exact search over the two shards == the single index's exact search: same ids, same fp64 distances
    assert two["verify"]["exact_digest"] == one["verify"]["exact_digest"] and two["verify"]["queries"] == 64
    assert two["verify"]["recall_at_k_vs_exact"] >= one["verify"]["recall_at_k_vs_exact"] - 0.05
    # value counts every sample of the one data set once
    assert abs(two["value"] - 6000 * two["steps"] / (two["ms_per_step"] * two["steps"] / 1e3)) / two["value"] < 1e-6


def test_bench_weak_scaling_is_still_available():
    one = _line([sys.executable, "bench.py", "--gpus", "1", "--scaling", "weak"] + SMALL[:-1])
    assert one["scaling"] == "weak" and one["config"]["samples_per_gpu"] == 6000
