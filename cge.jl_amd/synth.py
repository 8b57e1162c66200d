"""Seeded synthetic ABCD-like graphs + embeddings for the parity tests and bench.py.

The reference's papers use ABCDGraphGenerator.jl, which is not available offline; this is an
own generator with the same ingredients: power-law community sizes (exponent beta), power-law
degrees (exponent gamma), a mixing fraction xi of each vertex's stubs wired outside its community,
configuration-model wiring, self-loops and multi-edges dropped, no isolated vertices, unit weights.
The embedding is "community centre + isotropic noise" so the alpha search is non-degenerate.

Outputs follow ``parseargs()`` conventions (1-based ids, Fortran-ordered matrices).
"""
from __future__ import annotations

import numpy as np


def _powerlaw_ints(rng, count, exponent, lo, hi):
    """`count` integers in [lo, hi] with P(k) ~ k^-exponent (inverse-CDF of the continuous law)."""
    u = rng.random(count)
    a = 1.0 - exponent
    x = (lo ** a + u * ((hi + 1) ** a - lo ** a)) ** (1.0 / a)
    return np.clip(np.floor(x).astype(np.int64), lo, hi)


def _pair_stubs(rng, stubs):
    """Random perfect matching of a stub list (configuration model)."""
    stubs = stubs[rng.permutation(len(stubs))]
    if len(stubs) % 2:
        stubs = stubs[:-1]
    return stubs[0::2], stubs[1::2]


def abcd_like(n, m, n_comm, d, seed=42, xi=0.2, beta=1.5, gamma=2.5, directed=False, centre_scale=2.0,
              noise_scale=0.5, shuffle_ids=True):
    """Returns dict(edges (m',2) int64 F-order 1-based, eweights, vweights, comm (n,1) int64 1-based,
    clusters list, embedding (n,d) F-order)."""
    rng = np.random.default_rng(seed)
    # community sizes: power law, rescaled to sum to n, every community >= 8 vertices
    raw = _powerlaw_ints(rng, n_comm, beta, 10, 1000).astype(np.float64)
    sizes = np.maximum(8, np.floor(raw * (n / raw.sum())).astype(np.int64))
    diff = n - sizes.sum()
    order = np.argsort(-sizes)
    k = 0
    while diff != 0:
        j = order[k % n_comm]
        step = 1 if diff > 0 else -1
        if sizes[j] + step >= 8:
            sizes[j] += step
            diff -= step
        k += 1
    comm0 = np.repeat(np.arange(n_comm, dtype=np.int64), sizes)  # community of vertex (ordered ids)
    # degrees: power law rescaled so that sum = 2m
    deg = _powerlaw_ints(rng, n, gamma, 2, max(8, int(np.sqrt(n)))).astype(np.float64)
    deg = np.maximum(1, np.round(deg * (2.0 * m / deg.sum()))).astype(np.int64)
    ext = rng.binomial(deg, xi)
    inte = deg - ext
    # internal stubs: pair inside each community (sort stubs by community, pair within segments)
    stub_v = np.repeat(np.arange(n, dtype=np.int64), inte)
    key = comm0[stub_v].astype(np.float64) + rng.random(len(stub_v))  # random order inside a community
    stub_v = stub_v[np.argsort(key, kind="stable")]
    seg = comm0[stub_v]
    a, b = stub_v[0:-1:2], stub_v[1::2]
    ok = seg[0:-1:2] == seg[1::2]  # drop the pairs that straddle two communities
    a, b = a[ok[: len(a)]], b[ok[: len(b)]]
    ea, eb = _pair_stubs(rng, np.repeat(np.arange(n, dtype=np.int64), ext))
    src = np.concatenate([a, ea])
    dst = np.concatenate([b, eb])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if not directed:
        lo, hi = np.minimum(src, dst), np.maximum(src, dst)
        src, dst = lo, hi
    code = np.unique(src * n + dst)
    src, dst = code // n, code % n
    # no isolated vertices (a zero weight gives lweight = 0 -> NaN at src/landmarks.jl:402)
    seen = np.zeros(n, dtype=bool)
    seen[src] = True
    seen[dst] = True
    iso = np.flatnonzero(~seen)
    if len(iso):
        starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        mate = starts[comm0[iso]] + (iso - starts[comm0[iso]] + 1) % sizes[comm0[iso]]
        src = np.concatenate([src, np.minimum(iso, mate)])
        dst = np.concatenate([dst, np.maximum(iso, mate)])
        code = np.unique(src * n + dst)
        src, dst = code // n, code % n
    if directed:  # random orientation
        flip = rng.random(len(src)) < 0.5
        src, dst = np.where(flip, dst, src), np.where(flip, src, dst)
    # embedding
    centres = rng.standard_normal((n_comm, d)) * centre_scale
    emb = centres[comm0] + rng.standard_normal((n, d)) * noise_scale
    if shuffle_ids:  # vertex ids carry no community information (as in example/10k.ecg)
        perm = rng.permutation(n)  # new id of old vertex v is perm[v]
        src, dst = perm[src], perm[dst]
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        comm0 = comm0[inv]
        emb = emb[inv]
        if not directed:
            lo, hi = np.minimum(src, dst), np.maximum(src, dst)
            src, dst = lo, hi
    o = np.lexsort((dst, src))
    src, dst = src[o], dst[o]
    edges = np.asfortranarray(np.stack([src + 1, dst + 1], axis=1).astype(np.int64))
    eweights = np.ones(len(src))
    vweights = np.zeros(n)
    np.add.at(vweights, src, 1.0)
    np.add.at(vweights, dst, 1.0)
    comm = np.asfortranarray((comm0 + 1).reshape(-1, 1))
    order = np.argsort(comm0, kind="stable")
    bounds = np.flatnonzero(np.diff(comm0[order])) + 1
    clusters = [g.astype(np.int64) + 1 for g in np.split(order, bounds)]
    return {"edges": edges, "eweights": eweights, "vweights": vweights, "comm": comm, "clusters": clusters,
            "embedding": np.asfortranarray(emb), "n": n, "m": len(src), "d": d, "C": n_comm}
