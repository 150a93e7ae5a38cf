"""Seeded "synthetic intropolis" of SURVEY.md section 8(d), as CSR arrays.

N samples in 64 latent clusters; J junction keys "chr{1..22} {start} {end}";
junction j is "on" in a Geometric(0.35)-sized random set of clusters; a sample
carries j with p = 0.6 when its cluster is on, else p = 0.002; coverage =
1 + Geometric(0.45) clipped to 500; sample ids ascending within a line.  Samples
are numbered so that a cluster's members are contiguous (external id =
position + 1), which lets the whole matrix be drawn as two sorted Bernoulli
processes over the J x N cell grid and merged without a sort.
"""
import numpy as np

SEED = 8675309
QUERY_SEED = 8675310
N_CLUSTERS = 64
P_IN, P_BG = 0.6, 0.002


def make_keys(rng, J):
    chrom = rng.integers(1, 23, size=J)
    start = rng.integers(10_000, 240_000_000, size=J)
    end = start + rng.integers(60, 500_000, size=J)
    return ["chr%d %d %d" % (c, s, e) for c, s, e in zip(chrom.tolist(), start.tolist(), end.tolist())]


def synthetic_intropolis(N, J=200_000, seed=SEED, batch=4000):
    """Returns dict(keys, row_ptr int64[J+1], samples int64[nnz] (external ids,
    1-based), cov int32[nnz], sample_count=N)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    keys = make_keys(rng, J)
    # cluster sizes: sample -> cluster uniform; members of a cluster are contiguous
    sizes = rng.multinomial(N, np.full(N_CLUSTERS, 1.0 / N_CLUSTERS))
    cstart = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_on = np.minimum(rng.geometric(0.35, size=J), N_CLUSTERS)
    rows, counts = [], np.zeros(J, np.int64)
    for j0 in range(0, J, batch):
        j1 = min(J, j0 + batch)
        B = j1 - j0
        # which clusters are on: the n_on[j] smallest of 64 random keys
        r = rng.random((B, N_CLUSTERS))
        thresh = np.sort(r, axis=1)[np.arange(B), n_on[j0:j1] - 1]
        on = r <= thresh[:, None]                                  # [B, 64]
        # in-cluster carriers: Bernoulli(0.6) over every (junction, on cluster, member)
        jj, cc = np.nonzero(on)                                    # sorted by (j, c)
        run_len = sizes[cc]
        run_base = (jj.astype(np.int64) * N) + cstart[cc]
        total = int(run_len.sum())
        run_off = np.concatenate([[0], np.cumsum(run_len)])[:-1]
        hit = rng.random(total, dtype=np.float32) < P_IN
        cell = np.repeat(run_base - run_off, run_len) + np.arange(total, dtype=np.int64)
        a_in = cell[hit]                                           # sorted cell indices (j_local*N + s)
        # background carriers: Bernoulli(0.002) over the whole grid as geometric gaps
        n_bg = int(B * N * P_BG * 1.2) + 1000
        pos = np.cumsum(rng.geometric(P_BG, size=n_bg).astype(np.int64)) - 1
        while pos[-1] < B * N:
            more = np.cumsum(rng.geometric(P_BG, size=n_bg // 4 + 1000).astype(np.int64)) + pos[-1]
            pos = np.concatenate([pos, more])
        pos = pos[pos < B * N]
        pj, ps = pos // N, pos % N
        pc = np.searchsorted(cstart, ps, side="right") - 1
        pos = pos[~on[pj, pc]]                                     # those cells belong to the other process
        # merge the two sorted, disjoint event lists
        idx_b = np.searchsorted(a_in, pos) + np.arange(len(pos))
        merged = np.empty(len(a_in) + len(pos), np.int64)
        mask = np.ones(len(merged), bool)
        mask[idx_b] = False
        merged[idx_b] = pos
        merged[mask] = a_in
        counts[j0:j1] = np.bincount(merged // N, minlength=B)
        rows.append((merged % N).astype(np.int32))
    samples = np.concatenate(rows).astype(np.int64) + 1
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cov = np.minimum(1 + rng.geometric(0.45, size=len(samples)), 500).astype(np.int32)
    return dict(keys=keys, row_ptr=row_ptr, samples=samples, cov=cov, sample_count=int(N))


def query_items(n_items, nq, seed=QUERY_SEED):
    """1000 rows drawn without replacement, issued as by-item queries
    (mirrors the reference's tests/all_gtex_pancreas_nns.bash:19)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.choice(n_items, size=min(nq, n_items), replace=False).astype(np.int32)
