"""Every BASELINE.json configuration at its FULL workload, against committed oracle fixtures (`pytest -m gpu`).

config 1 (example/10k.* --force-exact) lives in test_gpu_parity.py::test_config1_force_exact_against_oracle_fixture.
Here: config 2 (100k / 1M, d = 64, -l 400 -m rss2), config 3 (1M / 20M, d = 128, -l 4000 -m diameter), config 4
(directed 1M, d = 128, --samples-local 1000000), the headline (1M / 10M, d = 128, -l 4000) and config 5 (d = 512,
-l 12000; at the vertex count stated in its test).  Fixtures: tests/golden/oracle_<name>.npz, produced offline by
tests/golden/make_oracle_fixture_cfg2.py / make_oracle_fixture_fullsize.py from the CPU oracle (the reference's O(n^2 d)
diameter loop replaced at 10^6 vertices by tests/diameter_ref.py, see there).  Graphs and sample draws are regenerated
here from their seeds and guarded by checksums.

Tolerances: integer outputs, unit-weight sums, centroids, d_ii and the diameter bit-exact; score vector and per-alpha
traces rtol 1e-9 (north star: 1e-6); Chung-Lu iteration counts identical."""
import math
import os
import sys
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, random_samples

pytestmark = pytest.mark.gpu
RTOL = 1e-9
sys.path.insert(0, ROOT)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def ctx():
    from cge.jl_amd import api

    c = api.Context(0)
    yield c
    c.close()


def _workload(name):
    import bench
    from cge.jl_amd import synth

    wl = bench.WORKLOADS[name]
    g = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"], seed=42, directed=bool(wl.get("directed", False)))
    return wl, g


def _fixture(name):
    path = os.path.join(GOLDEN, f"oracle_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    return np.load(path, allow_pickle=False)


def _check_landmarks(ctx, fx):
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm = ctx.landmarks_fetch()
    assert len(dii) == int(fx["N"]) and len(lw) == int(fx["n_ledges"])
    if "v_to_l" in fx:
        assert np.array_equal(v2l, fx["v_to_l"]), "v_to_l (raw landmark ids) differ from the oracle"
    else:
        assert crc(v2l.astype(np.int32)) == int(fx["v_to_l_crc"]), "v_to_l (raw landmark ids) differ from the oracle"
        assert np.array_equal(np.bincount(v2l)[1:], fx["landmark_sizes"]) and np.array_equal(v2l[::997], fx["v_to_l_sample"])
    assert np.array_equal(lcomm[:, 0], fx["lcomm"]) and np.array_equal(lweight, fx["lweight"])
    assert np.array_equal(dii, fx["dii"]), f"dii max diff {np.abs(dii - fx['dii']).max()}"
    assert crc(lemb) == int(fx["lemb_crc"]) and crc(ledges) == int(fx["ledges_crc"]) and crc(lw) == int(fx["lw_crc"])
    return lm


def _check_sweep(res, tr, fx, same_samples):
    exp = fx["result"]
    k = min(tr["n_alpha"], len(fx["iters"]))
    assert k >= 5 and tr["iters"][:k] == fx["iters"][:k].tolist(), "Chung-Lu iteration counts differ"
    d_got, d_exp = np.array(tr["div"][:k]), fx["div"][:k]
    both = np.isfinite(d_got) & np.isfinite(d_exp)
    assert both.sum() >= 5 and np.allclose(d_got[both], d_exp[both], rtol=RTOL, atol=0)
    assert res[0] == exp[0] and np.allclose(res[1:4], exp[1:4], rtol=RTOL, atol=1e-15)
    if same_samples:
        assert tr["n_alpha"] == len(fx["iters"])
        assert np.allclose(tr["auc"], fx["auc"], rtol=RTOL, atol=1e-12, equal_nan=True)
        assert res[4] == exp[4] and np.allclose(res[4:], exp[4:], rtol=RTOL, atol=1e-12)
    else:  # another stream of draws: the same distribution
        assert abs(res[5] - exp[5]) < 4 * (res[6] + exp[6]) + 1e-3
        S = None
    return k


def _run_config(ctx, name, check_host_flow=True):
    fx = _fixture(name)
    wl, g = _workload(name)
    directed = bool(wl.get("directed", False))
    assert g["n"] == int(fx["n"]) and g["m"] == int(fx["m"])
    assert crc(g["edges"]) == int(fx["edges_crc"]) and crc(g["embedding"]) == int(fx["emb_crc"]), \
        "the synthetic generator no longer reproduces the graph the fixture was made from"
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    ctx.set_option("diameter", 0)
    # (1) the device-resident flow of bench.py (library-drawn samples)
    res = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], directed=directed, seed=42,
                    auc_samples=wl["samples"])
    tr = ctx.last_trace
    hi, path, _, _ = ctx.last_diameter()
    assert hi == float(fx["hi"]), f"diameter {hi!r} ({path}) != exact CPU value {float(fx['hi'])!r}"
    lm = _check_landmarks(ctx, fx)
    _check_sweep(res, tr, fx, same_samples=False)
    assert res[6] == pytest.approx(1.96 * math.sqrt(res[5] * (1 - res[5]) / wl["samples"]), rel=1e-9)
    res_again = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], directed=directed, seed=42,
                          auc_samples=wl["samples"])
    assert np.array_equal(res, res_again)  # bitwise reproducible (fixed summation orders; unit weights)
    if check_host_flow:
        # (2) the reference's call shape with the fixture's sample draws: the whole vector and every trace
        dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
        smp = random_samples(np.random.default_rng(42), g["m"], g["n"], wl["samples"])
        res2 = ctx.wgcl(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, None, None, None, False,
                        auc_samples=wl["samples"], directed=directed, samples=smp, use_resident_original=True)
        _check_sweep(res2, ctx.last_trace, fx, same_samples=True)
    return wl, g, res, lm


def test_config2_rss2_full_size_against_oracle_fixture(ctx):
    """configs[1]: ABCD 100k vertices / 1M edges, d = 64, -l 400 -m rss2 (groups of up to ~6000 rows through
    rss2_walk_kernel) -- every output of landmarks() and wGCL() against the oracle's full-size run."""
    fx = _fixture("cfg2")
    wl, g, res, lm = _run_config(ctx, "cfg2")
    assert float(fx["hi"]) == ctx.last_diameter()[0]  # here the fixture's diameter is the oracle's own O(n^2 d) loop


def test_headline_full_size_against_oracle_fixture(ctx):
    """BASELINE.json metric workload: 10^6 vertices, ~10^7 edges, d = 128, -l 4000 -m rss."""
    _run_config(ctx, "headline")


def test_config3_diameter_rule_full_size_against_oracle_fixture(ctx):
    """configs[2]: 10^6 vertices / 2*10^7 edges, d = 128, -l 4000 -m diameter (about 30 dependent batches of splits)."""
    wl, g, res, lm = _run_config(ctx, "cfg3")
    # the cluster-pair scatter of the 2*10^7-edge list vs numpy
    C = g["C"]
    comm = g["comm"][:, 0]
    _, vc = ctx.edge_scatter(None, 1, C, False, want_wedges=False)
    ca, cb = comm[g["edges"][:, 0] - 1] - 1, comm[g["edges"][:, 1] - 1] - 1
    lo, hi_c = np.minimum(ca, cb), np.maximum(ca, cb)
    assert np.array_equal(vc, np.bincount(C * lo - lo * (lo - 1) // 2 + (hi_c - lo), minlength=C * (C + 1) // 2).astype(float))
    # pruned == brute force (5 * 10^11 pairs on the fp64 MFMA kernel)
    hi = ctx.last_diameter()[0]
    ctx.set_option("diameter", 1)
    try:
        res_b = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], seed=42, auc_samples=wl["samples"])
        hi_b, path_b, _, _ = ctx.last_diameter()
    finally:
        ctx.set_option("diameter", 0)
    assert path_b == "brute" and hi_b == hi and np.array_equal(res, res_b)


def test_config4_directed_million_samples_against_oracle_fixture(ctx):
    """configs[3]: directed 10^6-vertex graph, d = 128, --samples-local 1000000 (wGCL_directed, the device sampler with
    rejection against 10^7 resident edges, the C^2 form of vect_C)."""
    wl, g, res, lm = _run_config(ctx, "cfg4")
    n, m, S, C = g["n"], g["m"], wl["samples"], g["C"]
    # directed C x C scatter vs numpy
    comm = g["comm"][:, 0]
    _, vc = ctx.edge_scatter(None, 1, C, True, want_wedges=False)
    ca, cb = comm[g["edges"][:, 0] - 1] - 1, comm[g["edges"][:, 1] - 1] - 1
    assert np.array_equal(vc, np.bincount(ca * C + cb, minlength=C * C).astype(float))
    # the 10^6 device-drawn pairs: in range, ordered pairs i != j, none of them an edge, positives uniform over the rows
    pos, ni, nj = ctx.draw_samples(42, S, directed=True)
    assert pos.min() >= 1 and pos.max() <= m and ni.min() >= 1 and ni.max() <= n and nj.min() >= 1 and nj.max() <= n
    assert np.all(ni != nj)
    code = (g["edges"][:, 0] - 1) * n + (g["edges"][:, 1] - 1)
    assert not np.isin((ni - 1) * n + (nj - 1), code).any()
    assert (ni < nj).mean() == pytest.approx(0.5, abs=0.01)  # ordered pairs, both orientations
    cnt = np.bincount((pos - 1) * 64 // m, minlength=64)
    assert ((cnt - S / 64) ** 2 / (S / 64)).sum() < 64 + 6 * math.sqrt(128)
    # the same draws handed in as host arrays (host preparation path) give the bits of the device-resident flow
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    res_h = ctx.wgcl(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, None, None, None, False,
                     auc_samples=S, directed=True, samples=(pos.reshape(1, -1), ni.reshape(1, -1), nj.reshape(1, -1)),
                     use_resident_original=True)
    assert np.array_equal(res, res_h)


def test_config5_d512_twelve_thousand_landmarks(ctx):
    """configs[4] (ABCD 10M / 200M, d = 512, -l 12000) at the shape the test budget allows: n = 200 000 vertices, 4.2M edges,
    1500 communities, d = 512, 12 000 landmarks, i.e. the code paths of that configuration (the batched eigen-solver for
    128 < d <= 512, the fp32-MFMA bound pass at K = 512, 12 000-landmark sweep on the launch-per-iteration fit) -- the
    full-size run is `bench.py --workload cfg5` (the 41 GB embedding is generated in HBM).  No oracle can do 12 000 splits
    at d = 512 in test time, so: the reference's invariants, the exact diameter against the CPU branch and bound,
    pruned == brute force, fp32 bound pass == fp64 bound pass, host upload == device-pointer upload, reproducible bits."""
    import torch

    from cge.jl_amd import synth
    from diameter_ref import exact_diameter

    n, d, C, land = 200_000, 512, 1500, 12000
    g = synth.abcd_like(n, 4_200_000, C, d, seed=42)
    m = g["m"]
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    res = ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)
    tr = ctx.last_trace
    hi, path, pairs, tiles = ctx.last_diameter()
    ref_hi, hi_i, hi_j, _ = exact_diameter(g["embedding"], g["comm"][:, 0])
    assert hi == ref_hi, (hi, ref_hi, path)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = ctx.landmarks_fetch()
    N = len(dii)
    assert N == land and v2l.min() == 1 and v2l.max() == N
    comm = g["comm"][:, 0]
    first = np.zeros(N + 1, dtype=np.int64)
    first[v2l] = comm
    assert np.array_equal(first[v2l], comm) and np.array_equal(lcomm[:, 0], first[1:])  # landmarks nest in communities
    assert lw.sum() == m and lweight.sum() == 2 * m and np.array_equal(np.bincount(v2l, weights=g["vweights"])[1:], lweight)
    for l in np.random.default_rng(1).integers(1, N + 1, 25):
        mem = np.flatnonzero(v2l == l)
        w = g["vweights"][mem]
        cen = (g["embedding"][mem] * w[:, None]).sum(0) / w.sum()
        assert np.allclose(lemb[l - 1], cen, rtol=1e-12, atol=1e-14)
        assert dii[l - 1] == pytest.approx(np.sqrt(((g["embedding"][mem] - lemb[l - 1]) ** 2).sum() / w.sum()), rel=1e-12)
    assert np.all(np.isfinite(res)) and 0.25 <= res[0] <= 10 and 0 < res[1] <= math.log(2) and 0 <= res[5] <= 1
    best = int(np.nanargmin(tr["div"]))
    assert res[1] == tr["div"][best] and res[0] == 0.25 * (best + 1) and all(it >= 1 for it in tr["iters"])
    assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
    try:  # bound pass in fp64, then brute force: the same diameter bits, the same score
        ctx.set_option("diameter_f32", 0)
        assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
        assert ctx.last_diameter()[0] == hi
        ctx.set_option("diameter", 1)
        assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
        assert ctx.last_diameter()[:2] == (hi, "brute")
    finally:
        ctx.set_option("diameter", 0)
        ctx.set_option("diameter_f32", 1)
    # the embedding handed over as a device pointer (row-major torch tensor), as bench.py --workload cfg5 does
    X = torch.from_numpy(np.ascontiguousarray(g["embedding"])).to("cuda:0")
    torch.cuda.synchronize()
    ctx.set_embedding_device(X.data_ptr(), n, d, row_major=True)
    del X
    assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
