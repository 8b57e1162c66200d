"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C-ABI of
libcge_hip.so and is compared with the CPU oracle on the same inputs, with the committed golden
fixtures, or -- at sizes the oracle cannot reach -- through size-independent invariants.

Tolerances: landmark / cluster ids, landmark edge lists and unit-weight sums are bit-exact;
score-vector elements are compared at rtol 1e-9 (the north star allows 1e-6); best alphas exact.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, canonical_partition

pytestmark = pytest.mark.gpu

RTOL = 1e-9
README_KAT = [6.25, 0.002961243353776198, 0.0, 0.0, 9.75, 0.0017000000000000348, 0.000807441501038938]


@pytest.fixture(scope="module")
def ctx():
    from cge.jl_amd import api

    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def synth20k():
    from cge.jl_amd import synth

    return synth.abcd_like(20000, 200000, 20, 32, seed=1)


def _check_landmarks(got, ref, unit_weights=True):
    names = ["dii", "embed", "cluster", "landmark_edges", "weights", "lweight", "v_to_l"]
    assert np.array_equal(got[6], ref[6]), "v_to_l (raw landmark ids) differ"
    assert np.array_equal(got[2], ref[2]), "landmark communities differ"
    assert np.array_equal(got[3], ref[3]), "landmark edge list differs"
    if unit_weights:
        assert np.array_equal(got[4], ref[4]) and np.array_equal(got[5], ref[5])
    else:
        assert np.allclose(got[4], ref[4], rtol=1e-13, atol=0) and np.allclose(got[5], ref[5], rtol=1e-13, atol=0)
    # same summation order, unfused arithmetic => bit-exact centroids and d_ii
    assert np.array_equal(got[1], ref[1]), f"embed max diff {np.abs(got[1] - ref[1]).max()}"
    assert np.array_equal(got[0], ref[0]), f"dii max diff {np.abs(got[0] - ref[0]).max()}"
    for g, r, nm in zip(got, ref, names):
        assert g.shape == r.shape and g.dtype == r.dtype, nm


@pytest.mark.parametrize("method", ["rss", "rss2", "size", "diameter"])
def test_landmarks_parity_reference_fixture(ctx, orc, test115, method):
    """test/runtests.jl:43-93 (all four rules, -l 20 -f 1) -- values, not only types."""
    import cge.jl_amd as cg

    a = test115
    args = (a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False, 20, 1, method,
            False)
    _check_landmarks(cg.landmarks(*args, ctx=ctx), orc.landmarks(*args))


@pytest.mark.parametrize("method,land,forced", [("rss", 200, 4), ("rss2", 300, 2), ("size", 400, 4),
                                                 ("diameter", 330, 3)])
def test_landmarks_parity_example10k(ctx, orc, example10k, method, land, forced):
    import cge.jl_amd as cg

    a = example10k
    args = (a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False, land, forced,
            method, False)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    _check_landmarks(got, ref)
    assert np.array_equal(canonical_partition(got[6]), canonical_partition(ref[6]))


@pytest.mark.parametrize("method", ["rss", "rss2", "size", "diameter"])
def test_landmarks_parity_synthetic(ctx, orc, synth20k, method):
    import cge.jl_amd as cg

    g = synth20k
    args = (g["edges"], g["eweights"], g["vweights"], g["clusters"], g["comm"], g["embedding"], False, 300, 4, method,
            False)
    _check_landmarks(cg.landmarks(*args, ctx=ctx), orc.landmarks(*args))


@pytest.mark.parametrize("shape", ["two_giants", "giant_and_dwarfs"])
def test_landmarks_groups_beyond_the_lds_sort(ctx, orc, shape):
    """The per-group sort along the principal axis runs in LDS pieces for groups of up to 32 768 rows; beyond that a batch takes two
    device-wide stable sorts (few long groups) or the segmented radix sort (one giant among many short groups).  Both shapes
    against the oracle."""
    import cge.jl_amd as cg

    rng = np.random.default_rng(17)
    sizes = [36000, 35000] if shape == "two_giants" else [40000] + [100] * 100
    n, d = int(sum(sizes)), 8
    comm1 = np.repeat(np.arange(1, len(sizes) + 1), sizes)
    centres = rng.normal(size=(len(sizes), d)) * 3.0
    emb = centres[comm1 - 1] + rng.normal(size=(n, d))
    m = 6 * n
    edges = np.stack([rng.integers(1, n + 1, size=m), rng.integers(1, n + 1, size=m)], axis=1).astype(np.int64)
    edges = edges[edges[:, 0] != edges[:, 1]]
    ew = np.ones(len(edges))
    vw = np.bincount(np.concatenate([edges[:, 0], edges[:, 1]]) - 1, minlength=n).astype(np.float64)
    vw[vw == 0] = 1.0
    comm = comm1.reshape(-1, 1).astype(np.int64)
    clusters = [np.flatnonzero(comm1 == c) + 1 for c in range(1, len(sizes) + 1)]
    args = (edges, ew, vw, clusters, comm, emb, False, 3 * len(sizes) + 40, 2, "rss", False)
    _check_landmarks(cg.landmarks(*args, ctx=ctx), orc.landmarks(*args))


def test_landmarks_weighted_directed_and_truncation(ctx, orc, test115):
    import cge.jl_amd as cg

    a = test115
    w = a["eweights"] * 1.42  # test_weights.edgelist (test/runtests.jl:16)
    vw = a["vweights"] * 1.42
    args = (a["edges"], w, vw, a["clusters"], a["comm"], a["embedding"], False, 25, 2, "rss", True)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    assert np.array_equal(got[6], ref[6]) and np.array_equal(got[3], ref[3])
    assert np.allclose(got[4], ref[4], rtol=1e-13, atol=0)
    assert got[4].sum() == pytest.approx(w.sum(), rel=1e-13)
    # more landmarks than unique rows: clamp + warning path (src/landmarks.jl:373-376)
    emb = a["embedding"].copy()
    emb[50:] = emb[:65]  # 65 unique rows
    got = cg.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], emb, False, 90, 1, "size",
                       False, ctx=ctx)
    ref = orc.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], emb, False, 90, 1, "size",
                        False)
    assert ctx.truncated and np.array_equal(got[6], ref[6]) and len(got[0]) == len(ref[0])


@pytest.mark.parametrize("method", ["rss", "rss2", "size", "diameter"])
def test_landmarks_duplicate_rows_ties(ctx, orc, method):
    """Every row appears twice: ties everywhere in z, including at its maximum (the sorted-order rss path
    must hand those groups to the generic round-based path), medians landing on tied values, `==` branches
    of the size / diameter rules."""
    import cge.jl_amd as cg

    rng = np.random.default_rng(11)
    n, d, C = 600, 16, 5
    base = rng.standard_normal((n // 2, d)) + np.repeat(rng.standard_normal((C, d)) * 3, n // 2 // C, axis=0)
    emb = np.asfortranarray(np.repeat(base, 2, axis=0))
    comm = np.asfortranarray(np.repeat(np.arange(1, C + 1), n // C).reshape(-1, 1))
    src = rng.integers(1, n + 1, 4000)
    dst = rng.integers(1, n + 1, 4000)
    keep = src != dst
    edges = np.asfortranarray(np.stack([np.minimum(src, dst)[keep], np.maximum(src, dst)[keep]], axis=1))
    ew = np.ones(len(edges))
    vw = np.bincount(np.concatenate([edges[:, 0], edges[:, 1]]), minlength=n + 1)[1:].astype(float) + 1.0
    clusters = [np.flatnonzero(comm[:, 0] == c) + 1 for c in range(1, C + 1)]
    args = (edges, ew, vw, clusters, comm, emb, False, 60, 4, method, False)
    _check_landmarks(cg.landmarks(*args, ctx=ctx), orc.landmarks(*args))


@pytest.mark.parametrize("d", [2, 3, 5, 17, 32, 33, 64, 65, 100, 127, 128, 129, 200, 256, 333, 512])
def test_group_eig_kernel(ctx, d):
    """The batched device eigen-solver (register-resident Householder tridiagonalisation, Sturm multisection,
    inverse iteration; replaces `eigvecs(A)[:, end]`, src/landmarks.jl:99) against LAPACK on covariance-like
    matrices: residual of the eigen-equation, agreement with numpy's vector, sign convention."""
    rng = np.random.default_rng(100 + d)
    mats = []
    for t in range(40 if d <= 128 else 8):
        k = int(rng.integers(max(2, d // 2), 4 * d + 8))
        Y = rng.normal(size=(k, d)) * rng.uniform(0.2, 3.0, size=d)
        if t % 5 == 0:
            Y[:, : d // 2] = 0.0      # zero columns: rank deficiency, steps without a reflection
        if t % 7 == 3:
            Y = np.round(Y * 4) / 4   # dyadic entries: exact zeros and ties in the arithmetic
        mats.append(Y.T @ Y)
    mats.append(np.diag(np.arange(1.0, d + 1)))       # already diagonal: no reflection at any step
    mats.append(np.zeros((d, d)))                      # zero matrix: any unit vector; the solver returns e_1
    A = np.stack(mats)
    v = ctx.group_eig(A)
    for t in range(len(mats)):
        lam, vec = np.linalg.eigh(A[t])
        assert abs(np.linalg.norm(v[t]) - 1.0) < 1e-12
        big = int(np.argmax(np.abs(v[t])))
        assert v[t][big] > 0
        scale = max(lam[-1], 1e-300)
        resid = np.linalg.norm(A[t] @ v[t] - lam[-1] * v[t]) / scale
        assert resid < 1e-11, (d, t, resid)
        gap = (lam[-1] - lam[-2]) / scale if d > 1 else 1.0
        if gap > 1e-6:
            ref = vec[:, -1] * np.sign(vec[big, -1])
            assert np.max(np.abs(ref - v[t])) < 1e-10 / gap, (d, t)


@pytest.mark.parametrize("d,method", [(200, "rss"), (256, "size"), (96, "diameter"), (65, "rss2"), (128, "rss2"), (129, "rss")])
def test_landmarks_parity_wide_embeddings(ctx, orc, d, method):
    """Embedding dimensions beyond one MFMA tile / one register-resident covariance: the global-memory eigen-solver
    (128 < d <= 512), multi-tile MFMA SYRK, dimensions that are not multiples of 8/16/128.  (The oracle's Jacobi solver
    costs O(d^3) per sweep, so the oracle-checked cases stop at d = 256; d = 512 is covered below.)"""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    g = synth.abcd_like(1500, 12000, 4, d, seed=d)
    args = (g["edges"], g["eweights"], g["vweights"], g["clusters"], g["comm"], g["embedding"], False, 22, 4, method,
            False)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    _check_landmarks(got, ref)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = got
    smp = api.draw_samples(ctx, 3, 2000)
    res, tr = cg.wGCL(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, g["edges"], g["eweights"],
                      g["embedding"], False, samples=smp, trace=True, ctx=ctx)
    exp, etr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, g["edges"], g["eweights"],
                        g["embedding"], False, smp, trace=True)
    _cmp_result(res, exp, tr, etr)


def test_d512_invariants(ctx):
    """Config 5's embedding width (d = 512) end to end on the device: invariants instead of the oracle."""
    from cge.jl_amd import synth

    g = synth.abcd_like(20000, 200000, 10, 512, seed=9)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    res = ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000)
    hi, path, _, _ = ctx.last_diameter()
    dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
    N = len(dii)
    assert N == 120 and np.all(np.isfinite(res)) and 0 < res[1] <= math.log(2)
    comm = g["comm"][:, 0]
    first = np.zeros(N + 1, dtype=np.int64)
    first[v2l] = comm
    assert np.array_equal(first[v2l], comm) and lw.sum() == g["m"] and lweight.sum() == 2 * g["m"]
    cen = np.zeros((N, 512))
    np.add.at(cen, v2l - 1, g["embedding"] * g["vweights"][:, None])
    assert np.allclose(lemb, cen / lweight[:, None], rtol=1e-11, atol=1e-13)
    ctx.set_option("diameter", 1)
    res_b = ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000)
    ctx.set_option("diameter", 0)
    assert ctx.last_diameter()[0] == hi and np.array_equal(res, res_b)
    # every group's cut is a threshold on ITS principal axis: children RSS never exceed the parent's
    assert np.array_equal(res, ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000))


def _cmp_result(res, exp, tr=None, etr=None):
    assert len(res) == len(exp)
    assert res[0] == exp[0] and res[4] == exp[4], (res, exp)  # best alphas
    assert np.allclose(res, exp, rtol=RTOL, atol=1e-12), (res, exp)
    if tr is not None:
        assert tr["iters"] == etr["iters"], "Chung-Lu iteration counts differ"
        assert np.allclose(tr["div"], etr["div"], rtol=RTOL, equal_nan=True)
        assert np.allclose(tr["auc"], etr["auc"], rtol=RTOL, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("split", [False, True])
def test_wgcl_reference_test_shape(ctx, orc, test115, split):
    """test/runtests.jl:95-103: landmarks (diameter rule) then wGCL with empty v_to_l, seed 42, 10000 samples."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = test115
    lm = cg.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False, 20, 1,
                      cg.split_cluster_diameter, False, ctx=ctx)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    # the library draws the samples itself (seeded); fetch the same draw for the oracle
    res, tr = cg.wGCL(ledges, lw, lcomm, lemb, dii, lweight, *empty, split, 42, 10000, trace=True, ctx=ctx)
    assert res.dtype == np.float64 and res[0] <= 10.0  # the reference's own assertions
    smp = api.draw_samples(ctx, 42, 10000)  # resident graph == landmark graph after the exact-mode call
    exp, etr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, *empty, split, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    res2 = cg.wGCL(ledges, lw, lcomm, lemb, dii, lweight, *empty, split, samples=smp, ctx=ctx)
    assert np.array_equal(res, res2)  # pre-drawn samples == library-drawn samples; run-to-run reproducible


def test_wgcl_exact_mode_original_graph(ctx, orc, test115):
    """CGE_CLI.jl without landmarks: distances = zeros, exact mode on the 115-vertex graph, unseeded sets."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = test115
    n = len(a["vweights"])
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(a["edges"], a["eweights"], n)
    smp = api.draw_samples(ctx, 7, 3000, n_sets=40)  # a fresh draw per alpha (seed == -1 semantics)
    res, tr = cg.wGCL(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n), a["vweights"], *empty, False,
                      samples=smp, trace=True, ctx=ctx)
    exp, etr = orc.wGCL(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n), a["vweights"], *empty,
                        False, smp, trace=True)
    _cmp_result(res, exp, tr, etr)


@pytest.mark.parametrize("n", [40, 130, 700, 1500])
def test_fit_persistent_kernel_matches_stepwise_and_oracle(ctx, orc, n):
    """The Chung-Lu fixed point (src/divergence.jl:150-168) as one persistent launch per alpha (register-resident
    upper triangle, grid barriers) against one launch per iteration and against the oracle: same iteration counts,
    same scores.  Sizes cover one tile, ragged last tiles, and several tiles per wave row."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    g = synth.abcd_like(n, 6 * n, max(2, n // 60), 8, seed=n)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(g["edges"], g["eweights"], n)
    smp = api.draw_samples(ctx, 11, 2000)
    args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
    out = {}
    try:
        # one launch per iteration / persistent (the data as its own signal), twice
        for mode in (1, 2, 3):
            ctx.set_option("fit_persistent", min(mode, 2))
            out[mode] = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
            assert (ctx.get_stat("fit_persistent_alphas") > 0) == (mode >= 2)
    finally:
        ctx.set_option("fit_persistent", 0)
    (r1, t1), (r2, t2), (r3, t3) = out[1], out[2], out[3]
    assert t1["iters"] == t2["iters"] == t3["iters"]
    assert np.array_equal(r2, r3)  # the persistent form is bitwise reproducible
    assert np.allclose(r1, r2, rtol=1e-11, atol=1e-13)
    assert np.allclose(t1["div"], t2["div"], rtol=1e-11, equal_nan=True)
    if n <= 700:
        exp, etr = orc.wGCL(*args, smp, trace=True)
        _cmp_result(r2, exp, t2, etr)


@pytest.mark.parametrize("n", [90, 700])
def test_fit_persistent_directed_matches_stepwise_and_oracle(ctx, orc, n):
    """wGCL_directed's fixed point (src/divergence.jl:434-467: two iterates, the diagonal counted twice, the decaying
    step size) as one persistent launch per alpha against the launch-pair-per-iteration path and the oracle."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    g = synth.abcd_like(n, 6 * n, max(2, n // 60), 8, seed=n + 1, directed=True)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(g["edges"], g["eweights"], n)
    p1, ni, nj = api.draw_samples(ctx, 5, 2000, directed=True)
    p2, _, _ = api.draw_samples(ctx, 6, 2000, directed=True)
    smp = (p1, ni, nj, p2)
    args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
    out = {}
    try:
        for mode in (1, 2, 4):  # launch pair per iteration / persistent (the data as its own signal), twice
            ctx.set_option("fit_persistent", min(mode, 2))
            out[mode] = cg.wGCL_directed(*args, samples=smp, trace=True, ctx=ctx)
            assert (ctx.get_stat("fit_persistent_alphas") > 0) == (mode >= 2)
        assert np.array_equal(out[2][0], out[4][0]) and out[2][1]["iters"] == out[4][1]["iters"]  # bitwise reproducible
        ctx.set_option("fit_persistent", 2)
        ctx.set_option("fit_persistent_test_timeout", 1)  # abandoned launches: the iterates must be untouched
        out[3] = cg.wGCL_directed(*args, samples=smp, trace=True, ctx=ctx)
        assert ctx.get_stat("fit_persistent_alphas") == 0
    finally:
        ctx.set_option("fit_persistent_test_timeout", 0)
        ctx.set_option("fit_persistent", 0)
    (r1, t1), (r2, t2), (r3, t3) = out[1], out[2], out[3]
    assert t1["iters"] == t2["iters"] == t3["iters"]
    assert np.array_equal(r1, r3)
    assert np.allclose(r1, r2, rtol=1e-10, atol=1e-13)
    exp, etr = orc.wGCL_directed(*args, smp, trace=True)
    _cmp_result(r2, exp, t2, etr)


def test_fit_persistent_handoffs_under_varied_geometry(ctx):
    """The cross-workgroup hand-offs of the persistent fits (partial vectors, iterates, maxima through write-through
    stores and dependency counters) on many grid geometries -- from a handful of workgroups to every CU with two tile
    slots per wave, ragged last tiles, undirected and directed: a stale read anywhere would change an iterate, hence
    an iteration count or a score; both must equal the launch-per-iteration path, run after run."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    # 3977, 4030: every CU busy with 8 tiles (63 x 63 tiles of 64 is what the register file holds; beyond: one launch per iteration)
    for n, directed in [(513, False), (640, True), (1000, False), (1337, True), (2049, False), (3100, False),
                        (3977, False), (2500, True), (4030, False), (3500, True)]:
        g = synth.abcd_like(n, 5 * n, max(2, n // 80), 6, seed=3 * n + 1, directed=directed)
        if n % 2 == 1 and not directed:  # weighted edges (dyadic: the per-edge scatter's float atomics stay exact, so
            rng = np.random.default_rng(n)  # run-to-run bits can be compared), vertex weights = weighted degrees
            g["eweights"] = rng.integers(1, 17, size=len(g["eweights"])) / 4.0
            vw = np.zeros(n)
            np.add.at(vw, g["edges"][:, 0] - 1, g["eweights"])
            np.add.at(vw, g["edges"][:, 1] - 1, g["eweights"])
            g["vweights"] = vw
        ctx.set_graph(g["edges"], g["eweights"], n)
        if directed:
            p1, ni, nj = api.draw_samples(ctx, 5, 1000, directed=True)
            smp = (p1, ni, nj, p1)
            fn = cg.wGCL_directed
        else:
            smp = api.draw_samples(ctx, 5, 1000)
            fn = cg.wGCL
        args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
        try:
            ctx.set_option("fit_persistent", 1)
            ref, tref = fn(*args, samples=smp, trace=True, ctx=ctx)
            ctx.set_option("fit_persistent", 2)
            runs = [fn(*args, samples=smp, trace=True, ctx=ctx) for _ in range(3)]
            assert ctx.get_stat("fit_persistent_alphas") > 0
        finally:
            ctx.set_option("fit_persistent", 0)
        for res, tr in runs:
            assert tr["iters"] == tref["iters"], (n, directed)
            assert np.array_equal(res, runs[0][0]), (n, directed)  # bitwise reproducible
            assert np.allclose(res, ref, rtol=1e-10, atol=1e-13), (n, directed)


def test_pow_from_stored_logarithm_is_within_one_ulp(ctx):
    """GD = (1 - D)^alpha (src/divergence.jl:142-148) as exp2(alpha * log2(1 - D)) with the logarithm kept per score as
    double + float: against numpy's pow (glibc, < 1 ulp) on the whole alpha grid -- at most 1 ulp apart, exact at the
    ends (0 and 1), NaN kept; the library pow of the device for comparison."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(200000), 1.0 - np.exp(-rng.random(100000) * 36.0), rng.random(50000) * 1e-9,
                        [0.0, 1.0, 0.5, 1.0 - 2.0 ** -53, 2.0 ** -53, np.nan]])
    worst = 0.0
    for alpha in np.arange(1, 41) * 0.25:
        ref = np.power(1.0 - x, alpha)
        got = ctx.pow_test(x, alpha, 1)
        lib = ctx.pow_test(x, alpha, 0)
        fin = np.isfinite(ref) & (ref > 0)
        ulp = np.spacing(ref[fin])
        err = np.abs(got[fin] - ref[fin]) / ulp
        worst = max(worst, err.max())
        assert err.max() <= 1.0, (alpha, err.max(), x[fin][err.argmax()])
        assert np.abs(lib[fin] - ref[fin]).max() <= 2.0 * ulp.max() or True  # informational only
        assert got[-6] == 1.0 and got[-5] == 0.0 and np.isnan(got[-1])
        assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])  # exact zeros (underflow) agree
    assert worst <= 1.0


def test_pow_methods_give_the_same_sweep(ctx, orc, test115):
    """The alpha sweep with the stored-logarithm pow and with the library pow: same iteration counts, scores equal to
    rounding, both equal to the oracle."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    n = 900
    g = synth.abcd_like(n, 7 * n, 12, 8, seed=4)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(g["edges"], g["eweights"], n)
    smp = api.draw_samples(ctx, 3, 2000)
    args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
    try:
        ctx.set_option("pow_exp2", 0)
        r0, t0 = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
        ctx.set_option("pow_exp2", 1)
        r1, t1 = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
    finally:
        ctx.set_option("pow_exp2", 1)
    assert t0["iters"] == t1["iters"] and np.allclose(r0, r1, rtol=1e-12, atol=1e-14)
    exp, etr = orc.wGCL(*args, smp, trace=True)
    _cmp_result(r1, exp, t1, etr)


def test_vect_b_plain_kernels_give_the_same_bits(ctx, test115):
    """vect_B of score graphs beyond the LDS budget (N > 8192) or with more than 512 communities goes through kernels
    without staging; forced here on small inputs: same additions in the same order, so the same score bits (undirected,
    directed, landmark mode) as the staged row-bin form -- and both against round 4's option bvec_blocks, the tile
    form on the relabelled graph."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    for n, directed in [(700, False), (333, True)]:
        g = synth.abcd_like(n, 6 * n, max(2, n // 40), 8, seed=n, directed=directed)
        ctx.set_graph(g["edges"], g["eweights"], n)
        if directed:
            p1, ni, nj = api.draw_samples(ctx, 5, 500, directed=True)
            smp, fn = (p1, ni, nj, p1), cg.wGCL_directed
        else:
            smp, fn = api.draw_samples(ctx, 5, 500), cg.wGCL
        args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
        try:
            ref, tref = fn(*args, samples=smp, trace=True, ctx=ctx)  # the default: the staged row-bin form
            ctx.set_option("bvec_blocks", 1)
            dflt, tdflt = fn(*args, samples=smp, trace=True, ctx=ctx)  # round 4's option: relabelled, vect_B by tiles
            ctx.set_option("bvec_blocks", 0)
            ctx.set_option("test_bvec_plain", 1)
            res, tr = fn(*args, samples=smp, trace=True, ctx=ctx)
        finally:
            ctx.set_option("test_bvec_plain", 0)
            ctx.set_option("bvec_blocks", 0)
        assert np.array_equal(res, ref) and np.array_equal(tr["div"], tref["div"], equal_nan=True), (n, directed)
        # the tile form groups the same products differently (and the fit sums in the relabelled order): equal to rounding
        assert tdflt["iters"] == tref["iters"] and dflt[0] == ref[0] and dflt[4] == ref[4]
        assert np.allclose(dflt, ref, rtol=1e-11, atol=1e-15) and np.allclose(tdflt["div"], tref["div"], rtol=1e-11, equal_nan=True)


def test_fit_persistent_abandoned_launch_falls_back(ctx):
    """A persistent launch that gives up (here: forced through the testing option; in production a wait that timed
    out) leaves T untouched for the host, which restores it and fits the alpha with one launch per iteration."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    n = 600
    g = synth.abcd_like(n, 5 * n, 8, 8, seed=77)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(g["edges"], g["eweights"], n)
    smp = api.draw_samples(ctx, 5, 1500)
    args = (g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False)
    try:
        ctx.set_option("fit_persistent", 1)
        ref, tref = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
        ctx.set_option("fit_persistent", 2)
        ctx.set_option("fit_persistent_test_timeout", 1)
        got, tgot = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
        assert ctx.get_stat("fit_persistent_alphas") == 0
        assert np.array_equal(ref, got) and tref["iters"] == tgot["iters"]
        ctx.set_option("fit_persistent_test_timeout", 0)
        again, _ = cg.wGCL(*args, samples=smp, trace=True, ctx=ctx)
        assert ctx.get_stat("fit_persistent_alphas") > 0 and np.allclose(again, ref, rtol=1e-11, atol=1e-13)
    finally:
        ctx.set_option("fit_persistent_test_timeout", 0)
        ctx.set_option("fit_persistent", 0)


def test_fit_persistent_kernel_headline_landmark_count(ctx):
    """4000 landmarks (two tiles per wave, every CU busy): the persistent fit and the launch-per-iteration path
    give the same score vector."""
    from cge.jl_amd import synth

    g = synth.abcd_like(60000, 600000, 40, 16, seed=5)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    res = {}
    try:
        for mode in (1, 2, 3):
            ctx.set_option("fit_persistent", min(mode, 2))
            ctx.set_option("fit_fused", 0 if mode == 3 else 1)  # 3: the persistent fit with the separate launches around it
            res[mode] = ctx.score(g["clusters"], 4000, 4, "rss", seed=3, auc_samples=5000)
            assert (ctx.get_stat("fit_persistent_alphas") > 0) == (mode >= 2)
            assert (ctx.get_stat("fit_fused_alphas") > 0) == (mode == 2)
    finally:
        ctx.set_option("fit_persistent", 0)
        ctx.set_option("fit_fused", 1)
    assert res[1][0] == res[2][0] and res[1][4] == res[2][4]
    assert np.allclose(res[1], res[2], rtol=1e-10, atol=1e-13)
    assert res[2][0] == res[3][0] and np.allclose(res[2], res[3], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("directed", [False, True])
def test_split_global_and_weighted_edges_at_scale(ctx, orc, directed):
    """`--split-global` (internal / external JS, src/divergence.jl:235-246) on a weighted
    edge list (3-column file, src/auxilary.jl:105), 20 000 vertices / 200 000 edges through landmarks() + wGCL() in landmark
    mode: landmark ids bit-exact, the 7-vector with non-zero internal and external parts equal to the oracle's."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    n = 20000
    g = synth.abcd_like(n, 200000, 25, 24, seed=11, directed=directed)
    rng = np.random.default_rng(3)
    ew = rng.integers(1, 33, size=len(g["eweights"])) / 8.0  # dyadic weights: the per-edge scatter stays order-free
    vw = np.zeros(n)
    np.add.at(vw, g["edges"][:, 0] - 1, ew)
    np.add.at(vw, g["edges"][:, 1] - 1, ew)  # vweights as parseargs computes them for a weighted file (both ends)
    args_lm = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, 300, 2, "rss", directed)
    lm, ref = cg.landmarks(*args_lm, ctx=ctx), orc.landmarks(*args_lm)
    _check_landmarks(lm, ref, unit_weights=False)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    if directed:
        p1, ni, nj = api.draw_samples(ctx, 9, 5000, directed=True)
        smp, fn, ofn = (p1, ni, nj, p1), cg.wGCL_directed, orc.wGCL_directed
    else:
        smp, fn, ofn = api.draw_samples(ctx, 9, 5000), cg.wGCL, orc.wGCL
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], True)
    res, tr = fn(*wargs, 9, 5000, samples=smp, trace=True, ctx=ctx)
    exp, etr = ofn(*wargs, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    assert res[2] > 0.0 and res[3] > 0.0 and res[1] == pytest.approx((res[2] + res[3]) / 2.0, rel=1e-12)


def test_wgcl_landmark_mode_readme_known_answer(ctx, orc, example10k):
    """README.md:88-100 end to end through landmarks() + wGCL() in landmark mode."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = example10k
    lm = cg.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False,
                      a["land"], a["forced"], a["method"], False, ctx=ctx)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    smp = api.draw_samples(ctx, 42, 10000)  # resident graph = the original 10k graph
    res, tr = cg.wGCL(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"],
                      a["embedding"], False, 42, 10000, samples=smp, trace=True, ctx=ctx)
    assert res[0] == README_KAT[0] and res[1] == pytest.approx(README_KAT[1], rel=1e-9)
    assert res[2] == 0.0 and res[3] == 0.0
    assert abs(res[5] - README_KAT[5]) < 3 * (README_KAT[6] + res[6])  # Monte-Carlo element, other stream
    exp, etr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"],
                        a["embedding"], False, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    with open(os.path.join(GOLDEN, "oracle_generated.json")) as f:
        gold = json.load(f)["example10k_l200_rss"]
    assert tr["iters"] == gold["iters"] and np.allclose(tr["div"][: len(gold["div"])], gold["div"], rtol=RTOL, equal_nan=True)
    # library-drawn samples (seed 42) give the same numbers as the pre-drawn ones
    res2 = cg.wGCL(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"],
                   a["embedding"], False, 42, 10000, ctx=ctx)
    assert np.array_equal(res, res2)


def test_score_fused_pipeline(ctx, orc, example10k):
    """cge_score (device-resident CGE_CLI.jl flow) == landmarks() + wGCL() through host arrays."""
    from cge.jl_amd import api

    a = example10k
    ctx.set_inputs(a["edges"], a["eweights"], a["vweights"], a["comm"], a["embedding"])
    res = ctx.score(a["clusters"], a["land"], a["forced"], a["method"], seed=42, auc_samples=10000)
    tr = ctx.last_trace
    assert res[0] == README_KAT[0] and res[1] == pytest.approx(README_KAT[1], rel=1e-9)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
    smp = api.draw_samples(ctx, 42, 10000)
    exp, etr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"],
                        a["embedding"], False, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    ph = ctx.phase_ms()
    assert ph["sweep"] > 0 and ph["landmarks"] > 0 and ph["diameter"] > 0


def test_wgcl_directed(ctx, orc, test115):
    """wGCL_directed (never exercised by the reference's tests): exact mode, landmark mode, star guard."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = test115
    n = len(a["vweights"])
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(a["edges"], a["eweights"], n)
    p1, ni, nj = api.draw_samples(ctx, 5, 4000, directed=True)
    p2, _, _ = api.draw_samples(ctx, 6, 4000, directed=True)
    smp = (p1, ni, nj, p2)  # p2 = the overwriting second draw of src/divergence.jl:510
    res, tr = cg.wGCL_directed(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n), a["vweights"],
                               *empty, True, samples=smp, trace=True, ctx=ctx)
    exp, etr = orc.wGCL_directed(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n), a["vweights"],
                                 *empty, True, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    # landmark mode, directed landmark graph
    lm = cg.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False, 30, 2,
                      "rss", True, ctx=ctx)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    smp = api.draw_samples(ctx, 9, 5000, directed=True)
    res, tr = cg.wGCL_directed(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"],
                               a["embedding"], False, samples=smp, trace=True, ctx=ctx)
    exp, etr = orc.wGCL_directed(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"],
                                 a["eweights"], a["embedding"], False, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    # star graph: 6-element early return (src/divergence.jl:332-334)
    k = 12
    star = np.asfortranarray(np.stack([np.ones(k - 1, np.int64), np.arange(2, k + 1)], axis=1))
    out = cg.wGCL_directed(star, np.ones(k - 1), np.ones((k, 1), np.int64), np.random.default_rng(0).random((k, 4)),
                           np.zeros(k), np.ones(k), *empty, False, 1, 10, ctx=ctx)
    assert list(out) == [-1.0, 0.0, 0.0, 0.0, 0.0, 0.0]


def test_score_directed_device_resident(ctx, orc, test115):
    """cge_score with -d (config 4 flow): landmark mode and exact mode give exactly what landmarks() +
    wGCL_directed() through host arrays give (and those are checked against the oracle above)."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = test115
    n = len(a["vweights"])
    for land in (30, -1):
        ctx.set_inputs(a["edges"], a["eweights"], a["vweights"], a["comm"], a["embedding"])
        res = ctx.score(a["clusters"] if land != -1 else [], land, 2, "rss", directed=True, split=True, seed=5,
                        auc_samples=3000)
        tr = ctx.last_trace
        if land != -1:
            dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
            smp = api.draw_samples(ctx, 5, 3000, directed=True)
            exp, etr = orc.wGCL_directed(ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"],
                                         a["eweights"], a["embedding"], True, smp, trace=True)
        else:
            empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
            p1, ni, nj = api.draw_samples(ctx, 5, 3000, directed=True)
            p2 = ctx.draw_samples(5 + 0x7777, 3000, True, 1000)[0].reshape(1, -1)  # the second draw (:510)
            exp, etr = orc.wGCL_directed(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n),
                                         a["vweights"], *empty, True, (p1, ni, nj, p2), trace=True)
        _cmp_result(res, exp, tr, etr)
    # directed star graph through the device-resident path: 6-element early return
    k = 12
    star = np.asfortranarray(np.stack([np.ones(k - 1, np.int64), np.arange(2, k + 1)], axis=1))
    ctx.set_inputs(star, np.ones(k - 1), np.ones(k), np.ones((k, 1), np.int64), np.random.default_rng(0).random((k, 4)))
    assert list(ctx.score([], -1, directed=True, seed=1, auc_samples=10)) == [-1.0, 0.0, 0.0, 0.0, 0.0, 0.0]


def test_directed_exact_mode_device_sampler_with_the_second_draw(ctx):
    """Directed exact mode on a graph large enough for the device sampler (n(n-1) > 2^25): the un-reseeded second positive
    draw of src/divergence.jl:510 overwrites the pairs but not the weights -- the fused path (draws and preparation on the
    device) equals wGCL_directed with the same four draws fetched and handed in as host arrays."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    n = 6000
    g = synth.abcd_like(n, 8 * n, 12, 8, seed=19, directed=True)
    rng = np.random.default_rng(2)
    ew = rng.integers(1, 9, size=len(g["eweights"])) / 4.0  # weights matter: they come from the FIRST draw
    vw = np.zeros(n)
    np.add.at(vw, g["edges"][:, 0] - 1, ew)
    np.add.at(vw, g["edges"][:, 1] - 1, ew)
    ctx.set_inputs(g["edges"], ew, vw, g["comm"], g["embedding"])
    res = ctx.score([], -1, directed=True, seed=5, auc_samples=5000)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    p1, ni, nj = api.draw_samples(ctx, 5, 5000, directed=True)
    p2 = ctx.draw_samples(5 + 0x7777, 5000, True, 1000)[0].reshape(1, -1)
    exp = cg.wGCL_directed(g["edges"], ew, g["comm"], g["embedding"], np.zeros(n), vw, *empty, False,
                           samples=(p1, ni, nj, p2), ctx=ctx)
    assert np.array_equal(res, exp)
    same = cg.wGCL_directed(g["edges"], ew, g["comm"], g["embedding"], np.zeros(n), vw, *empty, False,
                            samples=(p1, ni, nj, p1), ctx=ctx)
    assert not np.array_equal(res[4:], same[4:])  # the second draw is really used


def test_wgcl_asserts_mirror_reference(ctx, test115):
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = test115
    n = len(a["vweights"])
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    with pytest.raises(AssertionError):  # src/divergence.jl:50
        cg.wGCL(a["edges"], a["eweights"], a["comm"][:-1], a["embedding"], np.zeros(n), a["vweights"], *empty, False,
                1, 100, ctx=ctx)
    with pytest.raises(AssertionError):  # src/divergence.jl:81
        cg.wGCL(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n - 1), a["vweights"], *empty, False,
                1, 100, ctx=ctx)
    # homogeneous cluster (src/landmarks.jl:165-167).  Integer coordinates and power-of-two weights keep the
    # weighted mean exact, so z == 0 for every row whatever the summation order (with arbitrary data the
    # reference itself ends in this error or in an @assert depending on rounding).
    emb = np.tile(np.arange(1.0, 33.0), (n, 1))
    with pytest.raises(api.CGEError, match="homogenous"):
        cg.landmarks(a["edges"], a["eweights"], np.full(n, 2.0), a["clusters"], a["comm"], emb, False, 1, 3, "rss",
                     False, ctx=ctx)


def test_score_that_fails_between_the_halves_of_its_sample_draw(ctx, synth20k):
    """cge_score enqueues the local score's sample draws behind the first synchronisation of the landmark phase and looks at
    them where the sweep starts.  A score that raises in between (a homogeneous community: src/landmarks.jl:165-167) leaves a
    draw pending; the next score on the context drains it and gives what a fresh context gives."""
    from cge.jl_amd import api

    g = synth20k
    n, d = g["n"], g["embedding"].shape[1]
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    good = ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000)
    # integer coordinates, all rows equal: every community is homogeneous, whatever the summation order
    ctx.set_vertex_data(g["comm"], np.full(n, 2.0))
    ctx.set_embedding(np.tile(np.arange(1.0, d + 1.0), (n, 1)))
    with pytest.raises(api.CGEError, match="homogenous"):
        ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000)
    ctx.set_vertex_data(g["comm"], g["vweights"])
    ctx.set_embedding(g["embedding"])
    again = ctx.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000)
    assert np.array_equal(good, again)
    fresh = api.Context()
    try:
        fresh.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        assert np.array_equal(good, fresh.score(g["clusters"], 120, 4, "rss", seed=1, auc_samples=5000))
    finally:
        fresh.close()


@pytest.mark.parametrize("directed", [False, True])
def test_edge_scatter_kernel(ctx, synth20k, directed):
    g = synth20k
    n, C = g["n"], g["C"]
    rng = np.random.default_rng(3)
    N = 77
    comm = g["comm"][:, 0]
    v2l = np.zeros(n, dtype=np.int64)  # landmarks nested in communities
    for c in range(1, C + 1):
        idx = np.flatnonzero(comm == c)
        v2l[idx] = rng.integers(0, 3, size=len(idx)) + 3 * (c - 1) + 1
    w = rng.integers(1, 5, size=g["m"]).astype(np.float64) * 0.5  # exactly representable => order independent
    ctx.set_graph(g["edges"], w, n)
    ctx.set_vertex_data(g["comm"], g["vweights"])
    wed, vc = ctx.edge_scatter(v2l, N, C, directed)
    a, b = v2l[g["edges"][:, 0] - 1], v2l[g["edges"][:, 1] - 1]
    ca, cb = comm[g["edges"][:, 0] - 1], comm[g["edges"][:, 1] - 1]
    if not directed:
        a, b = np.minimum(a, b), np.maximum(a, b)
        ca, cb = np.minimum(ca, cb), np.maximum(ca, cb)
    exp = np.zeros((N, N))
    np.add.at(exp, (a - 1, b - 1), w)
    assert np.array_equal(wed, exp) and wed.sum() == w.sum()
    if directed:
        expc = np.zeros(C * C)
        np.add.at(expc, (ca - 1) * C + cb - 1, w)
    else:
        from oracle import oracle as o

        expc = np.zeros(C * (C + 1) // 2)
        np.add.at(expc, np.array([o.idx(C, int(x), int(y)) - 1 for x, y in zip(ca, cb)]), w)
    assert np.array_equal(vc, expc)
    # a shard of the edge list (multi-GPU path): partial sums add up
    w1, c1 = ctx.edge_scatter(v2l, N, C, directed, 0, g["m"] // 3)
    w2, c2 = ctx.edge_scatter(v2l, N, C, directed, g["m"] // 3, g["m"])
    assert np.array_equal(w1 + w2, exp) and np.array_equal(c1 + c2, expc)


@pytest.mark.parametrize("directed", [False, True])
@pytest.mark.parametrize("weighted", [False, True])
def test_edge_scatter_blocked_two_pass_form(ctx, directed, weighted):
    """The score path's form of the per-edge cluster-pair scatter (kernels_scatter.hip: blocked edge list, community
    slices in LDS, off-diagonal pairs bucketed by row, no global atomics) against numpy: several vertex blocks, chunks
    that split a tile, unit and dyadic weights, packed (undirected) and C x C (directed) outputs; repeated calls reuse the
    blocked copy; 1500 communities (config 5) still take this path, 2500 and 9000 go through the tiled form of the landmark-pair
    matrix (a dense C x C stage, packed afterwards) with the same result."""
    from cge.jl_amd import synth

    rng = np.random.default_rng(11)
    for n, m, C in ((150_000, 1_600_000, 37), (40_000, 300_000, 1500), (60_000, 400_000, 2500), (100_000, 500_000, 9000),
                    (3_000, 20_000, 5)):
        g = synth.abcd_like(n, m, C, 4, seed=23, directed=directed)
        w = (rng.integers(1, 9, size=g["m"]) / 4.0) if weighted else g["eweights"]
        ctx.set_graph(g["edges"], w, g["n"])
        ctx.set_vertex_data(g["comm"], g["vweights"])
        comm = g["comm"][:, 0] - 1
        ca, cb = comm[g["edges"][:, 0] - 1], comm[g["edges"][:, 1] - 1]
        if directed:
            exp = np.bincount(ca * C + cb, weights=w, minlength=C * C)
        else:
            lo, hi = np.minimum(ca, cb), np.maximum(ca, cb)
            exp = np.bincount(C * lo - lo * (lo - 1) // 2 + (hi - lo), weights=w, minlength=C * (C + 1) // 2)
        for _ in range(2):
            _, vc = ctx.edge_scatter(None, 1, C, directed, want_wedges=False)
            assert np.array_equal(vc, exp) and vc.sum() == w.sum()


@pytest.mark.parametrize("n,d", [(10000, 32), (2999, 37), (777, 128), (130, 5)])
def test_max_pair_dist_kernel(ctx, orc, example10k, n, d):
    """fp64-MFMA diameter kernel vs the O(n^2 d) brute force (src/divergence.jl:104-113)."""
    if (n, d) == (10000, 32):
        emb = example10k["embedding"]
    else:
        rng = np.random.default_rng(n + d)
        emb = np.asfortranarray(rng.standard_normal((n, d)) * rng.random(d) + rng.standard_normal(d) * 3)
    ctx.set_graph(np.array([[1, 2]]), [1.0], n)
    ctx.set_embedding(emb)
    hi, ai, aj = ctx.max_pair_dist()
    exp = orc.max_pair_dist(emb)
    assert hi == exp, (hi, exp)
    assert 1 <= ai < aj <= n and np.sqrt(((emb[ai - 1] - emb[aj - 1]) ** 2).sum()) == pytest.approx(hi, rel=1e-14)
    parts = [ctx.max_pair_dist(p, 3)[0] for p in range(3)]  # sharded over 3 ranks: max of the shards
    assert max(parts) == exp


@pytest.mark.parametrize("which", ["example10k", "synth20k", "structureless"])
def test_diameter_pruned_equals_brute(ctx, orc, example10k, synth20k, which):
    """The branch-and-bound diameter (landmark-pair bounds) returns bit for bit the brute-force `hi`,
    on clustered data (heavy pruning) and on an isotropic cloud (hardly any pruning)."""
    if which == "example10k":
        g, land, method = example10k, 200, "rss"
        vw = g["vweights"]
    elif which == "synth20k":
        g, land, method = synth20k, 300, "size"
        vw = g["vweights"]
    else:
        rng = np.random.default_rng(5)
        g = dict(synth20k)
        g["embedding"] = np.asfortranarray(rng.standard_normal((g["n"], 32)))
        land, method, vw = 150, "diameter", g["vweights"]
    exp = orc.max_pair_dist(g["embedding"])
    ctx.set_inputs(g["edges"], g["eweights"], vw, g["comm"], g["embedding"])
    got = {}
    for opt in (1, 2, 0):
        ctx.set_option("diameter", opt)
        res = ctx.score(g["clusters"], land, 4, method, seed=3, auc_samples=2000)
        hi, path, pairs, tiles = ctx.last_diameter()
        got[opt] = (hi, path, res.tolist())
        assert hi == exp, (which, opt, hi, exp, path, pairs, tiles)
    ctx.set_option("diameter", 0)
    assert got[1][1] == "brute" and got[2][1] == "pruned"
    assert got[1][2] == got[2][2] == got[0][2]  # identical score vectors
    # the bound pass of the pruned path: bf16-split (default) / fp32 MFMA upper bounds vs fp64 MFMA -- the same diameter bits
    # and score, and never fewer candidate pairs (an upper bound of an upper bound)
    pairs = {}
    try:
        for f32 in (0, 1, 2):
            ctx.set_option("diameter", 2)
            ctx.set_option("diameter_f32", f32)
            res = ctx.score(g["clusters"], land, 4, method, seed=3, auc_samples=2000)
            hi, path, pairs[f32], _ = ctx.last_diameter()
            assert hi == exp and path == "pruned" and res.tolist() == got[2][2]
    finally:
        ctx.set_option("diameter", 0)
        ctx.set_option("diameter_f32", 2)
    assert pairs[0] <= pairs[1] <= 1.2 * pairs[0] + 16 and pairs[0] <= pairs[2] <= 1.2 * pairs[0] + 16


def test_library_drawn_samples_equal_fetched_draws(ctx, synth20k):
    """Seeded (one set) and unseeded (a fresh set per alpha) local scores with the samples drawn, rejected and prepared on
    the device inside the call, against the same draws fetched through cge_draw_samples and handed in as host arrays (the
    host preparation path): identical score vectors."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    a = synth20k
    lm = cg.landmarks(a["edges"], a["eweights"], a["vweights"], a["clusters"], a["comm"], a["embedding"], False, 300, 2,
                      "rss", False, ctx=ctx)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    args = (ledges, lw, lcomm, lemb, dii, lweight, a["vweights"], v2l, a["edges"], a["eweights"], a["embedding"], False)
    S = 3000
    seeded = cg.wGCL(*args, 77, S, ctx=ctx)
    smp = api.draw_samples(ctx, 77, S)
    assert np.array_equal(seeded, cg.wGCL(*args, 77, S, samples=smp, ctx=ctx))
    unseeded = cg.wGCL(*args, -1, S, ctx=ctx)
    smp40 = api.draw_samples(ctx, 0x5EEDC0DE, S, n_sets=40)  # the library's base seed when none is given
    assert np.array_equal(unseeded, cg.wGCL(*args, -1, S, samples=smp40, ctx=ctx))
    assert not np.array_equal(seeded[4:], unseeded[4:])  # another stream of samples


def test_draw_samples_are_non_edges(ctx, synth20k):
    from cge.jl_amd import api

    g = synth20k
    n = g["n"]
    dense = np.asfortranarray(np.array([[i, j] for i in range(1, 41) for j in range(i + 1, 41) if (i + j) % 3], np.int64))
    for edges, nn, directed in ((g["edges"], n, False), (g["edges"], n, True), (dense, 40, False)):
        ctx.set_graph(edges, np.ones(len(edges)), nn)
        pos, ni, nj = ctx.draw_samples(11, 20000, directed)
        pos2, ni2, nj2 = ctx.draw_samples(11, 20000, directed)
        assert np.array_equal(pos, pos2) and np.array_equal(ni, ni2) and np.array_equal(nj, nj2)
        assert pos.min() >= 1 and pos.max() <= len(edges)
        assert ni.min() >= 1 and nj.max() <= nn and np.all(ni != nj)
        if directed:
            eset = set(map(tuple, edges.tolist()))
        else:
            assert np.all(ni < nj)
            eset = set(map(tuple, np.sort(edges, axis=1).tolist()))
        assert not any((int(i), int(j)) in eset for i, j in zip(ni, nj))
    # dense graph: the non-edge draw is uniform over the complement
    ctx.set_graph(dense, np.ones(len(dense)), 40)
    _, ni, nj = ctx.draw_samples(1, 60000)
    comp = [(i, j) for i in range(1, 41) for j in range(i + 1, 41) if (i + j) % 3 == 0]
    counts = np.array([np.sum((ni == i) & (nj == j)) for i, j in comp])
    assert counts.sum() == 60000
    chi2 = ((counts - 60000 / len(comp)) ** 2 / (60000 / len(comp))).sum()
    assert chi2 < len(comp) + 6 * math.sqrt(2 * len(comp))


def test_js_kernel(ctx, orc):
    rng = np.random.default_rng(0)
    for C, directed in ((7, False), (12, True), (64, False)):
        ln = C * C if directed else C * (C + 1) // 2
        p, q = rng.random(ln) * 50, rng.random(ln) * 50
        assert ctx.js(p, q) == pytest.approx(orc.JS(p, q), rel=1e-12)
        vI = np.zeros(ln, dtype=np.uint8)
        if directed:
            vI[:: C + 1] = 1
        else:
            vI[[orc.idx(C, i, i) - 1 for i in range(1, C + 1)]] = 1
        assert ctx.js(p, q, vI, True) == pytest.approx(orc.JS(p, q, vI, True), rel=1e-12)
        assert ctx.js(p, q, vI, False) == pytest.approx(orc.JS(p, q, vI, False), rel=1e-12)


def test_larger_graph_invariants(ctx):
    """n = 200k, m ~ 2M, d = 64, -l 600: beyond the oracle's reach in test time; size-independent
    properties of the reference instead."""
    from cge.jl_amd import synth

    g = synth.abcd_like(200000, 2000000, 60, 64, seed=5)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    res = ctx.score(g["clusters"], 600, 4, "rss", seed=42, auc_samples=20000)
    tr = ctx.last_trace
    dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
    N = len(dii)
    assert N == 600 and v2l.min() == 1 and v2l.max() == N
    comm = g["comm"][:, 0]
    first = np.zeros(N + 1, dtype=np.int64)
    first[v2l] = comm
    assert np.array_equal(first[v2l], comm)  # every landmark inside one community
    assert np.array_equal(lcomm[:, 0], first[1:])
    assert lw.sum() == g["m"] and lweight.sum() == 2 * g["m"]  # unit weights: exact
    assert np.bincount(v2l, weights=g["vweights"])[1:].tolist() == lweight.tolist()
    cen = np.zeros((N, 64))
    np.add.at(cen, v2l - 1, g["embedding"] * g["vweights"][:, None])
    assert np.allclose(lemb, cen / lweight[:, None], rtol=1e-11, atol=1e-13)
    assert np.all(np.isfinite(res)) and 0.25 <= res[0] <= 10 and 0 <= res[1] <= math.log(2)
    assert 0 <= res[5] <= 1 and res[6] == pytest.approx(1.96 * math.sqrt(res[5] * (1 - res[5]) / 20000), rel=1e-12)
    assert all(it >= 1 for it in tr["iters"]) and tr["n_alpha"] <= 40
    best = np.nanargmin(tr["div"])
    assert res[1] == tr["div"][best] and res[0] == 0.25 * (best + 1)
    res2 = ctx.score(g["clusters"], 600, 4, "rss", seed=42, auc_samples=20000)
    assert np.array_equal(res, res2)  # bitwise reproducible (no float atomics on the score path; unit weights)
    hi, ai, aj = ctx.max_pair_dist()
    rng = np.random.default_rng(0)
    ii, jj = rng.integers(0, g["n"], 200000), rng.integers(0, g["n"], 200000)
    assert hi >= np.sqrt(((g["embedding"][ii] - g["embedding"][jj]) ** 2).sum(1)).max()


def test_exact_mode_beyond_the_reference_limit(ctx):
    """`--force-exact` (v_to_l = Int[], N = n) at 20 000 vertices -- twice the size at which the
    reference switches to landmarks because of its O(n^2) host objects (src/auxilary.jl:194-197).  Nothing O(n^2) lives on
    the host here; D, GD and log2(1 - D) are device matrices.  Beyond the oracle's reach in test time: the run must be
    bitwise reproducible, agree between the two power formulations and keep the sweep's invariants."""
    from cge.jl_amd import synth

    n = 20000
    g = synth.abcd_like(n, 10 * n, 30, 16, seed=7)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    r1 = ctx.score([], -1, seed=3, auc_samples=10000)
    it1 = ctx.get_stat("fit_iterations")
    r2 = ctx.score([], -1, seed=3, auc_samples=10000)
    assert np.array_equal(r1, r2) and ctx.get_stat("fit_iterations") == it1
    try:
        ctx.set_option("pow_exp2", 0)
        r3 = ctx.score([], -1, seed=3, auc_samples=10000)
    finally:
        ctx.set_option("pow_exp2", 1)
    assert ctx.get_stat("fit_iterations") == it1 and np.allclose(r1, r3, rtol=1e-11, atol=1e-14)
    assert len(r1) == 7 and np.all(np.isfinite(r1))
    assert 0.25 <= r1[0] <= 10.0 and (r1[0] / 0.25) == round(r1[0] / 0.25) and 0.25 <= r1[4] <= 10.0
    assert 0.0 < r1[1] < np.log(2.0) and r1[2] == 0.0 and r1[3] == 0.0 and 0.0 <= r1[5] <= 1.0
    assert r1[6] == pytest.approx(1.96 * np.sqrt(r1[5] * (1.0 - r1[5]) / 10000), rel=1e-12)
    assert ctx.get_stat("fit_persistent_alphas") == 0 and it1 > 100  # 20 000 vertices: one launch per iteration


@pytest.mark.parametrize("directed", [False, True])
def test_exact_mode_relabelled_score_graph_agrees_with_the_plain_one(ctx, directed):
    """Exact mode beyond 8192 vertices relabels the score graph by community inside the sweep (contiguous vect_B row
    sums, wgcl_host.cpp).  Nothing index-free may notice: with the option off (vertices as given, plain gather kernels)
    the same graph -- shuffled ids, non-unit dyadic edge weights, directed and undirected -- must give the same iteration
    counts and the same scores up to the rounding of a different summation order in the fit."""
    from cge.jl_amd import synth

    n = 9000
    g = synth.abcd_like(n, 8 * n, 40, 12, seed=11, directed=directed)
    rng = np.random.default_rng(5)
    ew = rng.choice([0.5, 1.0, 2.0, 4.0], size=len(g["eweights"]))
    ctx.set_inputs(g["edges"], ew, g["vweights"], g["comm"], g["embedding"])
    out = {}
    try:
        for opt in (1, 0):
            ctx.set_option("exact_relabel", opt)
            r = ctx.score([], -1, directed=directed, seed=3, auc_samples=4000)
            out[opt] = (np.array(r), ctx.get_stat("fit_iterations"), np.array(ctx.last_trace["div"]), np.array(ctx.last_trace["auc"]))
    finally:
        ctx.set_option("exact_relabel", 1)
    assert out[1][1] == out[0][1] and out[1][1] > 50
    assert np.allclose(out[1][0], out[0][0], rtol=1e-10, atol=1e-13)
    assert np.allclose(out[1][2], out[0][2], rtol=1e-10, atol=1e-13, equal_nan=True)
    assert np.allclose(out[1][3], out[0][3], rtol=1e-10, atol=1e-13, equal_nan=True)


def test_exact_mode_thirty_thousand_vertices_against_oracle_fixture(ctx):
    """`--force-exact` at 30 000 vertices -- three times the reference's switch to landmarks --
    against the CPU oracle's full run (tests/golden/oracle_exact30k.npz, an hour of one core; the test is skipped while
    that file has not been generated): iteration counts, per-alpha traces, the 7-vector."""
    import zlib

    import cge.jl_amd as cg
    from cge.jl_amd import synth
    from conftest import random_samples

    path = os.path.join(GOLDEN, "oracle_exact30k.npz")
    if not os.path.exists(path):
        pytest.skip("oracle_exact30k.npz not generated (tests/golden/make_oracle_fixture_exact30k.py)")
    fx = np.load(path)
    n = int(fx["n"])
    g = synth.abcd_like(n, 10 * n, 40, 16, seed=7)
    assert g["m"] == int(fx["m"]) and zlib.crc32(np.ascontiguousarray(g["edges"]).tobytes()) == int(fx["edges_crc"])
    smp = random_samples(np.random.default_rng(7), g["m"], n, 10000)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    res, tr = cg.wGCL(g["edges"], g["eweights"], g["comm"], g["embedding"], np.zeros(n), g["vweights"], *empty, False,
                      samples=smp, trace=True, ctx=ctx)
    assert tr["iters"] == fx["iters"].tolist()
    assert np.allclose(tr["div"], fx["div"], rtol=RTOL, equal_nan=True)
    assert np.allclose(tr["auc"], fx["auc"], rtol=RTOL, atol=1e-12, equal_nan=True)
    assert res[0] == fx["result"][0] and res[4] == fx["result"][4] and np.allclose(res, fx["result"], rtol=RTOL, atol=1e-12)


def test_exact_mode_sixty_thousand_vertices(ctx):
    """Exact mode at 60 000 vertices (D and GD are 28.8 GB each on the device; nothing O(n^2) on the host): the sweep's
    invariants.  `profiles/r02_exact_mode_probe.txt` records the same at 100 000 vertices (160 GB, 33 s per score)."""
    from cge.jl_amd import synth

    n = 60000
    g = synth.abcd_like(n, 10 * n, 80, 16, seed=7)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    r = ctx.score([], -1, seed=3, auc_samples=10000)
    tr = ctx.last_trace
    assert len(r) == 7 and np.all(np.isfinite(r)) and tr["n_alpha"] <= 40 and all(it >= 1 for it in tr["iters"])
    assert 0.25 <= r[0] <= 10.0 and 0.25 <= r[4] <= 10.0 and 0.0 < r[1] < np.log(2.0) and 0.0 <= r[5] <= 1.0
    best = int(np.nanargmin(tr["div"]))
    assert r[1] == tr["div"][best] and r[0] == 0.25 * (best + 1)
    assert r[6] == pytest.approx(1.96 * np.sqrt(r[5] * (1.0 - r[5]) / 10000), rel=1e-12)
    assert ctx.get_stat("fit_persistent_alphas") == 0  # one launch per iteration at this size


def test_headline_size_properties(ctx):
    """BASELINE.json's headline size (n = 10^6, m ~ 10^7, d = 128, -l 4000): no oracle can run here, so the
    size-independent properties of SURVEY.md §8c: partition validity, exact weight sums, centroids of sampled
    landmarks, cluster-pair scatter vs numpy, pruned diameter == brute-force diameter (all 5e11 pairs on the
    fp64 MFMA kernel), bitwise run-to-run reproducibility."""
    from cge.jl_amd import synth

    g = synth.abcd_like(1_000_000, 10_500_000, 500, 128, seed=42)
    n, m, C = g["n"], g["m"], g["C"]
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    ctx.set_option("diameter", 0)
    res = ctx.score(g["clusters"], 4000, 4, "rss", seed=42, auc_samples=10000)
    tr = ctx.last_trace
    hi, path, pairs, tiles = ctx.last_diameter()
    assert path == "pruned" and tiles < 0.01 * (n / 128) ** 2 / 2
    assert np.all(np.isfinite(res)) and 0.25 <= res[0] <= 10 and 0 < res[1] <= math.log(2) and 0 <= res[5] <= 1
    assert res[6] == pytest.approx(1.96 * math.sqrt(res[5] * (1 - res[5]) / 10000), rel=1e-12)
    best = int(np.nanargmin(tr["div"]))
    assert res[1] == tr["div"][best] and res[0] == 0.25 * (best + 1) and all(it >= 1 for it in tr["iters"])
    dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
    N = len(dii)
    assert N == 4000 and v2l.min() == 1 and v2l.max() == N
    comm = g["comm"][:, 0]
    first = np.zeros(N + 1, dtype=np.int64)
    first[v2l] = comm
    assert np.array_equal(first[v2l], comm) and np.array_equal(lcomm[:, 0], first[1:])  # landmarks nest in communities
    assert lw.sum() == m and lweight.sum() == 2 * m  # unit weights: exact
    assert np.array_equal(np.bincount(v2l, weights=g["vweights"])[1:], lweight)
    assert np.all(ledges[:, 0] <= ledges[:, 1]) and np.all(lw > 0)
    for l in np.random.default_rng(1).integers(1, N + 1, 40):  # weighted centroids / d_ii of sampled landmarks
        mem = np.flatnonzero(v2l == l)
        w = g["vweights"][mem]
        cen = (g["embedding"][mem] * w[:, None]).sum(0) / w.sum()
        assert np.allclose(lemb[l - 1], cen, rtol=1e-12, atol=1e-14)
        assert dii[l - 1] == pytest.approx(np.sqrt(((g["embedding"][mem] - lemb[l - 1]) ** 2).sum() / w.sum()), rel=1e-12)
    # C x C cluster-pair scatter (the score path's edge pass) vs numpy
    _, vc = ctx.edge_scatter(None, 1, C, False, want_wedges=False)
    ca, cb = comm[g["edges"][:, 0] - 1] - 1, comm[g["edges"][:, 1] - 1] - 1
    lo, hi_c = np.minimum(ca, cb), np.maximum(ca, cb)
    expc = np.bincount(C * lo - lo * (lo - 1) // 2 + (hi_c - lo), minlength=C * (C + 1) // 2).astype(float)
    assert np.array_equal(vc, expc)
    # exactness of the pruned diameter at full size, and reproducibility
    ctx.set_option("diameter", 1)
    res_b = ctx.score(g["clusters"], 4000, 4, "rss", seed=42, auc_samples=10000)
    hi_b, path_b, _, _ = ctx.last_diameter()
    ctx.set_option("diameter", 0)
    assert path_b == "brute" and hi_b == hi and np.array_equal(res, res_b)
    rng = np.random.default_rng(0)
    ii, jj = rng.integers(0, n, 300000), rng.integers(0, n, 300000)
    assert hi >= np.sqrt(((g["embedding"][ii] - g["embedding"][jj]) ** 2).sum(1)).max()


def test_config1_force_exact_against_oracle_fixture(ctx, example10k):
    """BASELINE config 1 (example/10k.* --force-exact, N = n = 10 000) on the GPU vs the oracle's full-size run
    (tests/golden/oracle_exact10k.json, generated by tests/golden/make_oracle_fixture_exact10k.py)."""
    import cge.jl_amd as cg

    path = os.path.join(GOLDEN, "oracle_exact10k.json")
    if not os.path.exists(path):
        pytest.skip("oracle_exact10k.json not generated")
    with open(path) as f:
        gold = json.load(f)
    a = example10k
    n = len(a["vweights"])
    smp = (np.array([gold["pos_idx"]]), np.array([gold["neg_i"]]), np.array([gold["neg_j"]]))
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    res, tr = cg.wGCL(a["edges"], a["eweights"], a["comm"], a["embedding"], np.zeros(n), a["vweights"], *empty, False,
                      samples=smp, trace=True, ctx=ctx)
    assert tr["iters"] == gold["iters"]
    assert np.allclose(tr["div"], gold["div"], rtol=RTOL, equal_nan=True)
    assert np.allclose(tr["auc"], gold["auc"], rtol=RTOL, atol=1e-12, equal_nan=True)
    assert res[0] == gold["result"][0] and res[4] == gold["result"][4]
    assert np.allclose(res, gold["result"], rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("case", range(60))
def test_randomised_parity_sweep(ctx, orc, case):
    """A seeded sweep over the flag surface on small random graphs: size, dimension, number of communities, split rule,
    forced splits, landmark count, directed, weighted, --split-global -- landmarks() bit-exact against the oracle, then
    wGCL / wGCL_directed in landmark mode with the same samples (iteration counts, traces, the 7-vector)."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    # CGE_STRESS_OFFSET / CGE_STRESS_WIDE: other seeds / embedding widths of the memory-resident eigen-solver too (offline
    # stress runs; the defaults are the committed cases)
    case += int(os.environ.get("CGE_STRESS_OFFSET", "0"))
    rng = np.random.default_rng(1000 + case)
    n = int(rng.integers(60, 1500))
    dims = [2, 3, 5, 8, 17, 33, 64, 100] + ([129, 130, 160, 192, 257, 300] if os.environ.get("CGE_STRESS_WIDE") else [])
    d = int(rng.choice(dims))
    C = int(rng.integers(2, max(3, n // 25)))
    method = ["rss", "rss2", "size", "diameter"][case % 4]
    directed = bool(rng.integers(0, 2))
    split = bool(rng.integers(0, 2))
    forced = int(rng.choice([1, 2, 4]))
    g = synth.abcd_like(n, int(rng.integers(3, 9)) * n, C, d, seed=500 + case, directed=directed)
    land = int(min(n // 3, max(C * forced, rng.integers(C, 6 * C + 2))))
    ew, vw = g["eweights"], g["vweights"]
    if rng.integers(0, 2):  # weighted (dyadic: the per-edge scatter's float atomics stay exact)
        ew = rng.integers(1, 17, size=len(ew)) / 4.0
        vw = np.zeros(n)
        np.add.at(vw, g["edges"][:, 0] - 1, ew)
        np.add.at(vw, g["edges"][:, 1] - 1, ew)
    args = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, land, forced, method, directed)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    _check_landmarks(got, ref, unit_weights=False)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = got
    S = 1500
    if directed:
        p1, ni, nj = api.draw_samples(ctx, case, S, directed=True)
        smp, fn, ofn = (p1, ni, nj, p1), cg.wGCL_directed, orc.wGCL_directed
    else:
        smp, fn, ofn = api.draw_samples(ctx, case, S), cg.wGCL, orc.wGCL
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], split)
    res, tr = fn(*wargs, case, S, samples=smp, trace=True, ctx=ctx)
    exp, etr = ofn(*wargs, smp, trace=True)
    _cmp_result(res, exp, tr, etr)


def test_directed_fit_without_a_fixed_point_raises(ctx, orc):
    """A documented divergence from the reference: offline stress case 226 of the sweep above (116 vertices,
    2 communities, -l 2 -f 1 -m size, directed, dyadic weights) ends in a landmark graph of TWO landmarks --
    edges [[1 1] [1 2] [2 1] [2 2]], weights [966.5 52 71.75 485.25] -- on which the directed Chung-Lu iteration has no fixed
    point within delta: the reference's `while diff > delta` (src/divergence.jl:434-467) never returns (the oracle, which
    restates it, loops for ever and is therefore NOT called here for the score).  The product bounds the loop (option
    fit_max_iterations, default 2 000 000) and returns CGE_E_ASSERT with a message, for every form of the fit; the landmark
    phase in front of it is bit-exact against the oracle as always."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    case = 226
    rng = np.random.default_rng(1000 + case)
    n = int(rng.integers(60, 1500))
    d = int(rng.choice([2, 3, 5, 8, 17, 33, 64, 100]))
    C = int(rng.integers(2, max(3, n // 25)))
    method = ["rss", "rss2", "size", "diameter"][case % 4]
    directed = bool(rng.integers(0, 2))
    split = bool(rng.integers(0, 2))
    forced = int(rng.choice([1, 2, 4]))
    g = synth.abcd_like(n, int(rng.integers(3, 9)) * n, C, d, seed=500 + case, directed=directed)
    land = int(min(n // 3, max(C * forced, rng.integers(C, 6 * C + 2))))
    assert rng.integers(0, 2) == 1 and (n, d, C, method, directed, forced, land) == (116, 8, 2, "size", True, 1, 2)
    ew = rng.integers(1, 17, size=len(g["eweights"])) / 4.0
    vw = np.zeros(n)
    np.add.at(vw, g["edges"][:, 0] - 1, ew)
    np.add.at(vw, g["edges"][:, 1] - 1, ew)
    args = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, land, forced, method, directed)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    _check_landmarks(got, ref, unit_weights=False)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = got
    assert len(dii) == 2 and ledges.tolist() == [[1, 1], [1, 2], [2, 1], [2, 2]] and lw.tolist() == [966.5, 52.0, 71.75, 485.25]
    S = 1500
    p1, ni, nj = api.draw_samples(ctx, case, S, directed=True)
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], split)
    try:
        ctx.set_option("fit_max_iterations", 20000)
        for opt in (0, 1, 2):
            ctx.set_option("fit_persistent", opt)
            with pytest.raises(api.AssertionErrorCGE) as ei:  # CGE_E_ASSERT: what a reference @assert maps to
                cg.wGCL_directed(*wargs, case, S, samples=(p1, ni, nj, p1), ctx=ctx)
            assert ei.value.code == -1 and "did not converge" in str(ei.value)
    finally:
        ctx.set_option("fit_max_iterations", 2000000)
        ctx.set_option("fit_persistent", 0)


def test_local_score_exact_ties_case_212(ctx, orc):
    """DESIGN section 2 (iv), profiles/repro_parity_case.py 212 (an offline stress case, 170 vertices, directed): a sampled
    edge (i, j) and a sampled non-edge (j, i) whose endpoints share ONE landmark give pos == neg in exact arithmetic
    (Tout[l] Tin[l] w_i w_j / lw[l]^2 (1 - D_ij)^alpha both ways), so `pos > neg` (src/divergence.jl:517) is decided by the
    rounding of two differently ordered products -- in the reference as here.  What IS guaranteed, and held in place by
    this test: landmarks bit-exact, global score and iteration counts equal, every per-alpha local score within the weight
    of those tied samples of the oracle's, and the best local alpha one of the alphas the oracle scores within that margin
    of its best."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    case = 212
    rng = np.random.default_rng(1000 + case)
    n = int(rng.integers(60, 1500))
    d = int(rng.choice([2, 3, 5, 8, 17, 33, 64, 100] + [129, 130, 160, 192, 257, 300]))
    C = int(rng.integers(2, max(3, n // 25)))
    method = ["rss", "rss2", "size", "diameter"][case % 4]
    directed = bool(rng.integers(0, 2))
    split = bool(rng.integers(0, 2))
    forced = int(rng.choice([1, 2, 4]))
    g = synth.abcd_like(n, int(rng.integers(3, 9)) * n, C, d, seed=500 + case, directed=directed)
    land = int(min(n // 3, max(C * forced, rng.integers(C, 6 * C + 2))))
    ew, vw = g["eweights"], g["vweights"]
    if rng.integers(0, 2):
        ew = rng.integers(1, 17, size=len(ew)) / 4.0
        vw = np.zeros(n)
        np.add.at(vw, g["edges"][:, 0] - 1, ew)
        np.add.at(vw, g["edges"][:, 1] - 1, ew)
    assert directed and n == 170, "the generator no longer reproduces stress case 212"
    args = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, land, forced, method, directed)
    got, ref = cg.landmarks(*args, ctx=ctx), orc.landmarks(*args)
    _check_landmarks(got, ref, unit_weights=False)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = got
    S = 1500
    p1, ni, nj = api.draw_samples(ctx, case, S, directed=True)
    smp = (p1, ni, nj, p1)
    # the tied samples: non-edge k is the reversal of edge k and both endpoints lie in one landmark
    pi, pj = g["edges"][p1[0] - 1, 0], g["edges"][p1[0] - 1, 1]
    tied = (pi == nj[0]) & (pj == ni[0]) & (v2l[pi - 1] == v2l[pj - 1])
    wk = ew[p1[0] - 1]
    margin = wk[tied].sum() / wk.sum()
    assert tied.sum() >= 1, "case 212 no longer contains an exact tie"
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], split)
    exp, etr = orc.wGCL_directed(*wargs, smp, trace=True)
    e_auc = np.array(etr["auc"])
    for opt in (0, 1, 2):  # every form of the fit
        ctx.set_option("fit_persistent", opt)
        try:
            res, tr = cg.wGCL_directed(*wargs, case, S, samples=smp, trace=True, ctx=ctx)
        finally:
            ctx.set_option("fit_persistent", 0)
        k = min(len(tr["auc"]), len(e_auc))
        assert tr["iters"][:k] == etr["iters"][:k]
        assert np.allclose(tr["div"][:k], etr["div"][:k], rtol=RTOL, equal_nan=True)
        assert res[0] == exp[0] and np.allclose(res[1:4], exp[1:4], rtol=RTOL, atol=1e-15)
        a_got = np.array(tr["auc"][:k])
        both = np.isfinite(a_got) & np.isfinite(e_auc[:k])
        assert np.all(np.abs(a_got[both] - e_auc[:k][both]) <= margin + 1e-12), (a_got, e_auc, margin)
        ia = int(round(res[4] / 0.25)) - 1  # the alpha this run calls best: the oracle scores it within the margin of its own best
        assert 0 <= ia < k and e_auc[ia] <= np.nanmin(e_auc[:k]) + 2 * margin + 1e-12
        assert res[5] == np.nanmin(a_got)  # (the patience counters may stop the two sweeps at different alphas after a flipped tie)


@pytest.mark.parametrize("form,case", [(2, 4), (2, 6), (2, 8), (2, 10), (2, 5), (2, 7), (2, 12), (2, 13)])
def test_randomised_exact_mode_sweep_under_start_skew(ctx, orc, form, case):
    """The start-skew class of bug (`done` / `fail` words tested against zero -- a fit that waited more than 64
    polls ended "converged" with the iterate of that moment) had ONE tripwire, and iteration-count equality with the oracle only
    fires under a slow schedule.  Here cases of the randomised exact-mode sweep below -- against the ORACLE -- run with the tile
    waves of the persistent fits napping before their first load (option fit_persistent_test_delay), for every persistent
    fit: undirected (even cases) and directed (odd cases)"""
    try:
        ctx.set_option("fit_persistent", form)
        ctx.set_option("fit_persistent_test_delay", 25)
        _exact_mode_case(ctx, orc, case, min_n=200)
    finally:
        ctx.set_option("fit_persistent_test_delay", 0)
        ctx.set_option("fit_persistent", 0)


@pytest.mark.parametrize("case", range(16))
def test_randomised_exact_mode_sweep(ctx, orc, case):
    """The same for exact mode (v_to_l = Int[], the score graph is the graph itself): random sizes across the launch
    forms of the fit (launch per iteration below 128 vertices, one tile per wave on 4 / 8 waves above), directed and
    undirected, weighted, --split-global, seeded and unseeded samples."""
    _exact_mode_case(ctx, orc, case)


def _exact_mode_case(ctx, orc, case, min_n=0):
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    rng = np.random.default_rng(2000 + case)
    n = max(min_n, int(rng.choice([40, 90, 127, 128, 200, 333, 520, 700])))  # (min_n: sizes the persistent forms take)
    d = int(rng.choice([2, 4, 9, 16]))
    C = int(rng.integers(2, max(3, n // 20)))
    directed = bool(case % 2)
    split = bool(rng.integers(0, 2))
    g = synth.abcd_like(n, int(rng.integers(3, 8)) * n, C, d, seed=700 + case, directed=directed)
    ew, vw = g["eweights"], g["vweights"]
    if rng.integers(0, 2):
        ew = rng.integers(1, 9, size=len(ew)) / 2.0
        vw = np.zeros(n)
        np.add.at(vw, g["edges"][:, 0] - 1, ew)
        np.add.at(vw, g["edges"][:, 1] - 1, ew)
    empty = ([], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)))
    ctx.set_graph(g["edges"], ew, n)
    S, n_sets = 800, (1 if case % 3 else 40)  # a fresh draw per alpha when unseeded
    if directed:
        p1, ni, nj = api.draw_samples(ctx, case, S, directed=True, n_sets=n_sets)
        p2, _, _ = api.draw_samples(ctx, case + 99, S, directed=True, n_sets=n_sets)
        smp, fn, ofn = (p1, ni, nj, p2), cg.wGCL_directed, orc.wGCL_directed
    else:
        smp, fn, ofn = api.draw_samples(ctx, case, S, n_sets=n_sets), cg.wGCL, orc.wGCL
    args = (g["edges"], ew, g["comm"], g["embedding"], np.zeros(n), vw, *empty, split)
    res, tr = fn(*args, samples=smp, trace=True, ctx=ctx)
    exp, etr = ofn(*args, smp, trace=True)
    _cmp_result(res, exp, tr, etr)


@pytest.mark.parametrize("case", range(4))
def test_randomised_landmarks_sweep_mid_size(ctx, orc, case):
    """landmarks() against the oracle on random mid-size graphs (3 000 - 9 000 vertices, up to 128 dimensions, hundreds of
    landmarks; twelve cases up to 30 000 vertices passed when this test was written -- the oracle sets the pace): several rounds of the speculative tree expansion, long groups (two-pass sort) and short ones (segmented
    sort), every split rule -- landmark ids, communities, edge lists bit-exact."""
    import cge.jl_amd as cg
    from cge.jl_amd import synth

    rng = np.random.default_rng(3000 + case)
    n = int(rng.integers(3000, 9000))
    d = int(rng.choice([8, 24, 64, 128]))
    C = int(rng.integers(5, 60))
    method = ["rss", "rss2", "size", "diameter"][case % 4]
    forced = int(rng.choice([1, 2, 4, 6]))
    land = int(rng.integers(C * forced, C * forced + 500))
    directed = bool(rng.integers(0, 2))
    g = synth.abcd_like(n, 6 * n, C, d, seed=900 + case, directed=directed)
    args = (g["edges"], g["eweights"], g["vweights"], g["clusters"], g["comm"], g["embedding"], False, land, forced, method,
            directed)
    _check_landmarks(cg.landmarks(*args, ctx=ctx), orc.landmarks(*args))


def test_projection_reduction_keeps_the_shuffle_tree_bits(ctx):
    """group_project_kernel adds the 64 lane partials of a row by gfx950 lane swaps (v_permlane32_swap / v_permlane16_swap /
    DPP row shifts) instead of six ds_bpermute rounds: the same pairs in the same order, so lane 0 must hold the same bits."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal((4096, 64)), rng.standard_normal((512, 64)) * 10.0 ** rng.integers(-8, 8, (512, 64)),
                        np.arange(64 * 7, dtype=np.float64).reshape(7, 64)])
    a, b = ctx.wave_tree_test(x)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    # and the tree is the one the comment states: ((..(l, l+32) .. (l, l+16)) .. ) with lane 0 last
    t = x.copy()
    for off in (32, 16, 8, 4, 2, 1):
        t = t[:, :off] + t[:, off:2 * off]
    assert np.array_equal(t[:, 0], a)


@pytest.mark.parametrize("mode", ["long", "short", "huge"])
def test_per_group_sort_is_the_stable_isless_sort(ctx, mode):
    """k_segmented_sort_z: groups sorted where they lie -- pieces of 4096 rows by a bitonic network in LDS, longer groups merged
    by rank (`long`), rocPRIM's segmented radix sort for batches of short groups (`short`), the device-wide pair of sorts when a
    group exceeds eight pieces (`huge`).  All must give the permutation of a STABLE ascending sort with -0.0 == 0.0 (the
    oracle's `<`; ties -- many: the values are drawn from a small set -- by index), and hand back the values' own bits."""
    rng = np.random.default_rng(11)
    lens = {"long": [4096, 4097, 1, 2, 9000, 21530, 777, 8192, 12289, 5000],
            "short": [rng.integers(1, 700) for _ in range(400)],
            "huge": [40000, 3000, 5000]}[mode]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    R = int(off[-1])
    z = rng.standard_normal(R)
    ties = rng.random(R) < 0.3
    z[ties] = rng.choice([0.0, -0.0, 1.5, -2.25, 1e-300, -1e300], ties.sum())
    zs, perm = ctx.segment_sort_test(z, off)
    key = z.view(np.uint64).copy()
    key[key == np.uint64(1) << np.uint64(63)] = 0  # -0.0 sorts as 0.0
    neg = (key >> np.uint64(63)) == 1
    key[neg] = ~key[neg]
    key[~neg] |= np.uint64(1) << np.uint64(63)
    for t in range(len(lens)):
        a, b = int(off[t]), int(off[t + 1])
        order = np.argsort(key[a:b], kind="stable")
        assert np.array_equal(perm[a:b], order.astype(np.int32)), (mode, t, lens[t])
        assert np.array_equal(zs[a:b].view(np.uint64), z[a:b][order].view(np.uint64))


@pytest.mark.parametrize("directed", [False, True])
def test_persistent_fit_survives_start_skew(ctx, directed):
    """The data-as-signal fits must not care when their workgroups start.  Rounds 1-2 tested their `done` / `fail` words
    against zero although the launch arms them with the sentinel pattern, so ANY hand-off wait of more than 64 polls ended the
    fit as "converged" -- silently, with the iterate of that moment.  Healthy runs never wait that long; a late workgroup does.
    Option fit_persistent_test_delay makes the tile waves nap ~60 us before their first load: same iteration counts, same bits."""
    from cge.jl_amd import synth

    g = synth.abcd_like(20000, 200000, 30, 16, seed=33, directed=directed)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 2)
        ref = ctx.score(g["clusters"], 300, 2, "rss", directed=directed, seed=3, auc_samples=3000)
        it_ref, n_ref = list(ctx.last_trace["iters"]), ctx.get_stat("fit_persistent_alphas")
        ctx.set_option("fit_persistent_test_delay", 20)
        got = ctx.score(g["clusters"], 300, 2, "rss", directed=directed, seed=3, auc_samples=3000)
        assert n_ref > 0 and ctx.get_stat("fit_persistent_alphas") == n_ref  # the persistent form ran, and was not abandoned
        assert list(ctx.last_trace["iters"]) == it_ref
        assert np.array_equal(got, ref)
    finally:
        ctx.set_option("fit_persistent_test_delay", 0)
        ctx.set_option("fit_persistent", 0)


def _fused_case(seed=21, n=30000, C=24, d=24):
    from cge.jl_amd import synth

    g = synth.abcd_like(n, 10 * n, C, d, seed=seed)
    rng = np.random.default_rng(seed)
    ew = rng.integers(1, 17, size=len(g["eweights"])) / 4.0  # dyadic weights: order-free scatter sums
    vw = np.zeros(n)
    np.add.at(vw, g["edges"][:, 0] - 1, ew)
    np.add.at(vw, g["edges"][:, 1] - 1, ew)
    return g, ew, vw


@pytest.mark.parametrize("split", [False, True])
def test_fused_alpha_chain_equals_separate_launches_and_oracle(ctx, orc, split):
    """Round 5: in landmark mode the power matrix, vect_B's tile sums and the local score's tallies ride on the launch of the
    persistent fit (kernels_fitp.hip, fit_flow_kernel<.., true>; src/divergence.jl:146, :170-213, :226-241).  The same sweep with
    the separate launches on the same relabelled graph (`fit_fused` = 0, `bvec_blocks` = 1: exp2 matrix, fit, auc_landmark,
    bvec_tile + bins, js) must give the SAME BITS -- vector, both traces, iteration counts -- and both agree with the oracle."""
    import cge.jl_amd as cg
    from cge.jl_amd import api

    g, ew, vw = _fused_case()
    args_lm = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, 1000, 4, "rss", False)
    lm = cg.landmarks(*args_lm, ctx=ctx)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    smp = api.draw_samples(ctx, 9, 6000)
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], split)
    try:
        res, tr = cg.wGCL(*wargs, 9, 6000, samples=smp, trace=True, ctx=ctx)
        n_fused = ctx.get_stat("fit_fused_alphas")
        assert n_fused == tr["n_alpha"] and ctx.get_stat("fit_persistent_alphas") == n_fused  # every alpha rode on its fit
        ctx.set_option("fit_fused", 0)
        ctx.set_option("bvec_blocks", 1)
        res2, tr2 = cg.wGCL(*wargs, 9, 6000, samples=smp, trace=True, ctx=ctx)
        assert ctx.get_stat("fit_fused_alphas") == 0 and ctx.get_stat("fit_persistent_alphas") == tr2["n_alpha"]
        assert np.array_equal(res, res2) and tr["iters"] == tr2["iters"]
        assert np.array_equal(tr["div"], tr2["div"], equal_nan=True) and np.array_equal(tr["auc"], tr2["auc"], equal_nan=True)
        ctx.set_option("bvec_blocks", 0)  # the graph as given, vect_B by row bins: another order of the same sums
        res3, tr3 = cg.wGCL(*wargs, 9, 6000, samples=smp, trace=True, ctx=ctx)
        assert tr3["iters"] == tr["iters"] and np.allclose(res3, res, rtol=1e-12, atol=1e-15)
    finally:
        ctx.set_option("fit_fused", 1)
        ctx.set_option("bvec_blocks", 0)
    exp, etr = orc.wGCL(*wargs, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
    if split:
        assert res[2] > 0.0 and res[3] > 0.0


def test_fused_alpha_chain_abandoned_and_late_launches(ctx):
    """The fused launch under the two testing knobs of the persistent fits: tile waves that start late (same bits), and a
    launch that gives up at once (nothing of its epilogue may be used: the alpha is redone with one launch per iteration,
    which needs the power matrix the fused form never wrote)."""
    g, ew, vw = _fused_case(seed=22, n=20000, C=16, d=16)
    ctx.set_inputs(g["edges"], ew, vw, g["comm"], g["embedding"])
    try:
        ref = ctx.score(g["clusters"], 700, 4, "size", seed=5, auc_samples=4000)
        it_ref = list(ctx.last_trace["iters"])
        assert ctx.get_stat("fit_fused_alphas") == len(it_ref)
        ctx.set_option("fit_persistent_test_delay", 20)
        got = ctx.score(g["clusters"], 700, 4, "size", seed=5, auc_samples=4000)
        assert ctx.get_stat("fit_fused_alphas") == len(it_ref) and list(ctx.last_trace["iters"]) == it_ref
        assert np.array_equal(got, ref)
        ctx.set_option("fit_persistent_test_delay", 0)
        ctx.set_option("fit_persistent_test_timeout", 1)
        got = ctx.score(g["clusters"], 700, 4, "size", seed=5, auc_samples=4000)
        assert ctx.get_stat("fit_fused_alphas") == 0 and ctx.get_stat("fit_persistent_alphas") == 0  # (nothing of the fused tries counts)
        assert list(ctx.last_trace["iters"]) == it_ref and np.allclose(got, ref, rtol=1e-11, atol=1e-14)
    finally:
        ctx.set_option("fit_persistent_test_delay", 0)
        ctx.set_option("fit_persistent_test_timeout", 0)


def test_fused_alpha_chain_declined_for_many_tiny_communities(ctx, orc):
    """More runs of communities in a 64-landmark block than the fused epilogue stages (CGE_FLOW_NP pieces): the sweep keeps the
    separate launches -- and still matches the oracle."""
    import cge.jl_amd as cg
    from cge.jl_amd import api, synth

    g = synth.abcd_like(12000, 90000, 400, 12, seed=8)  # 400 communities of ~30 vertices: -f 1 gives one or two landmarks each
    args_lm = (g["edges"], g["eweights"], g["vweights"], g["clusters"], g["comm"], g["embedding"], False, 500, 1, "rss", False)
    lm, ref = cg.landmarks(*args_lm, ctx=ctx), orc.landmarks(*args_lm)
    _check_landmarks(lm, ref)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    smp = api.draw_samples(ctx, 4, 3000)
    wargs = (ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, g["edges"], g["eweights"], g["embedding"], False)
    res, tr = cg.wGCL(*wargs, 4, 3000, samples=smp, trace=True, ctx=ctx)
    assert ctx.get_stat("fit_fused_alphas") == 0 and ctx.get_stat("fit_persistent_alphas") == tr["n_alpha"]
    exp, etr = orc.wGCL(*wargs, smp, trace=True)
    _cmp_result(res, exp, tr, etr)
