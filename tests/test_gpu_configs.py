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


_GRAPHS = {}  # the last generated graph (config 3 and its `size` twin share one: 75 s of numpy each otherwise)


def _workload(name):
    import bench
    from cge.jl_amd import synth

    wl = bench.WORKLOADS[name]
    key = (wl["n"], wl["m"], wl["C"], wl["d"], bool(wl.get("directed", False)))
    if key not in _GRAPHS:
        _GRAPHS.clear()
        _GRAPHS[key] = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"], seed=42, directed=key[4])
    return wl, _GRAPHS[key]


def _fixture(name):
    path = os.path.join(GOLDEN, f"oracle_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    return np.load(path, allow_pickle=False)


class _Prefixed:
    """A view of the keys `<prefix><name>` of a fixture that holds several runs (oracle_d512.npz)."""

    def __init__(self, fx, prefix):
        self.fx, self.prefix = fx, prefix

    def _key(self, k):
        return self.prefix + k if (self.prefix + k) in self.fx else k

    def __contains__(self, k):
        return self._key(k) in self.fx

    def __getitem__(self, k):
        return self.fx[self._key(k)]


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


def test_config2_graph_d128_rss2_long_groups_against_oracle_fixture(ctx):
    """config 2's graph with a 128-wide embedding under rss2 (oracle_cfg2_d128.npz): groups of >= 5000 rows through the
    64 < d <= 128 form of the chain kernel (rss2_chain_lds_kernel<2,4>: one row buffer, many rounds) -- src/landmarks.jl:92-147."""
    from cge.jl_amd import synth

    fx = _fixture("cfg2_d128")
    g = synth.abcd_like(100_000, 1_050_000, 50, 128, seed=42)
    assert g["n"] == int(fx["n"]) and g["m"] == int(fx["m"]) and crc(g["edges"]) == int(fx["edges_crc"]) and \
        crc(g["embedding"]) == int(fx["emb_crc"]), "the synthetic generator no longer reproduces the fixture's graph"
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    ctx.set_option("diameter", 0)
    res = ctx.score(g["clusters"], 400, 4, "rss2", seed=42, auc_samples=10000)
    tr = ctx.last_trace
    assert ctx.last_diameter()[0] == float(fx["hi"])
    lm = _check_landmarks(ctx, fx)
    assert max(len(c) for c in g["clusters"]) >= 5000  # the first splits of such a community are chains over thousands of rows
    _check_sweep(res, tr, fx, same_samples=False)
    assert np.array_equal(res, ctx.score(g["clusters"], 400, 4, "rss2", seed=42, auc_samples=10000))
    smp = random_samples(np.random.default_rng(42), g["m"], g["n"], 10000)
    res_h = ctx.wgcl(lm[3], lm[4], lm[2], lm[1], lm[0], lm[5], g["vweights"], lm[6], None, None, None, False, auc_samples=10000,
                     directed=False, samples=smp, use_resident_original=True)
    _check_sweep(res_h, ctx.last_trace, fx, same_samples=True)


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


def test_config3_graph_size_rule_against_oracle_fixture(ctx):
    """The `size` rule (median cuts, src/landmarks.jl:212-238) at full size: config 3's graph (10^6 vertices, 2*10^7 edges,
    d = 128, -l 4000) with -m size -- the one split rule the BASELINE configurations themselves do not use."""
    _run_config(ctx, "cfg3_size", check_host_flow=False)


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


@pytest.mark.parametrize("which", ["oracle_d512", "oracle_d512_lowrank", "oracle_d512_quick"])
@pytest.mark.parametrize("method", ["rss", "diameter"])
def test_d512_against_oracle_fixture(ctx, method, which):
    """The oracle pins of config 5's code paths (tests/golden/make_oracle_fixture_d512.py), d = 512: `oracle_d512` (20 000
    vertices, 30 communities, -l 300 -f 4; eigenvectors from LAPACK's syevr as in the reference), `_lowrank` (6000 vertices in
    20 communities of ~300 rows, -l 200: EVERY covariance rank-deficient, ~120 splits in the global phase; LAPACK) and `_quick`
    (12 000 vertices, 6 communities, -l 36; the oracle's own Jacobi) -- group_eig_panel_kernel (128 < d <= 512), the tile-pair
    covariance and the K = 512 fp32-MFMA bound pass against the CPU oracle: v_to_l, d_ii, weights and communities bit for bit,
    centroid / landmark-edge checksums, the diameter's bits (the oracle's O(n^2 d) loop), iteration counts, the 7-vector and
    every trace at 1e-9, for the rss rule and for a cut rule."""
    from cge.jl_amd import synth

    path = os.path.join(GOLDEN, which + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    fx0 = np.load(path, allow_pickle=False)
    if method + "_N" not in fx0:
        pytest.skip(f"{path} holds no {method} run yet (the generator saves after every rule)")
    fx = _Prefixed(fx0, method + "_")
    gen = (20_000, 210_000, 30, 300) if "gen_n" not in fx0 else tuple(int(fx0[k]) for k in ("gen_n", "gen_m", "gen_C", "land"))
    land, forced = gen[3], (int(fx0["forced"]) if "forced" in fx0 else 4)
    g = synth.abcd_like(gen[0], gen[1], gen[2], 512, seed=42)
    assert g["n"] == int(fx["n"]) and g["m"] == int(fx["m"])
    assert crc(g["edges"]) == int(fx["edges_crc"]) and crc(g["embedding"]) == int(fx["emb_crc"])
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    ctx.set_option("diameter", 0)
    res = ctx.score(g["clusters"], land, forced, method, seed=42, auc_samples=10000)
    tr = ctx.last_trace
    hi, dpath, _, _ = ctx.last_diameter()
    assert hi == float(fx["hi"]) and dpath == "pruned", (hi, float(fx["hi"]), dpath)
    lm = _check_landmarks(ctx, fx)
    _check_sweep(res, tr, fx, same_samples=False)
    dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
    smp = random_samples(np.random.default_rng(42), g["m"], g["n"], 10000)
    res2 = ctx.wgcl(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, None, None, None, False, auc_samples=10000,
                    samples=smp, use_resident_original=True)
    _check_sweep(res2, ctx.last_trace, fx, same_samples=True)
    for opt, val, back in (("diameter_f32", 0, 2), ("diameter_f32", 1, 2), ("diameter", 1, 0)):  # fp64 / fp32 bound pass, brute force
        try:
            ctx.set_option(opt, val)
            assert np.array_equal(res, ctx.score(g["clusters"], land, forced, method, seed=42, auc_samples=10000))
            assert ctx.last_diameter()[0] == hi
        finally:
            ctx.set_option(opt, back)


def test_config5_full_size_ten_million_vertices():
    """configs[4] AT ITS WORKLOAD: ABCD-like 10^7 vertices / 2*10^8 edges, d = 512, 1500 communities, -l 12000 -f 4 -m rss, the
    41 GB embedding generated in HBM and handed over as a device pointer (bench.py --workload cfg5).  No CPU oracle can run
    this (12 000 Jacobi problems at d = 512; 10^14 pair distances); its code paths are pinned by
    test_d512_against_oracle_fixture.  Here: the reference's invariants at full size, the landmark statistics of sampled
    landmarks against numpy, and the diameter two ways -- from the landmark partition on the device, and by the CPU branch and
    bound on a 10^5-row sample that contains the arg-max pair: the same bits; a second score in a fresh context: the same bits."""
    import torch

    import bench
    from cge.jl_amd import api, synth
    from diameter_ref import exact_diameter

    wl = bench.WORKLOADS["cfg5"]
    n, d, C, land = wl["n"], wl["d"], wl["C"], wl["land"]
    free, _ = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~190 GB of free HBM")
    g = synth.abcd_like(n, int(wl["m"] * 1.05), C, 1, seed=42)
    m = g["m"]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    centres = torch.randn(C, d, generator=gen, device=dev, dtype=torch.float64) * 2.0
    comm_dev = torch.from_numpy(g["comm"][:, 0] - 1).to(dev)
    X = torch.empty(n, d, dtype=torch.float64, device=dev)
    for a in range(0, n, 1 << 20):
        b = min(n, a + (1 << 20))
        X[a:b] = centres[comm_dev[a:b]] + torch.randn(b - a, d, generator=gen, device=dev, dtype=torch.float64) * 0.5
    torch.cuda.synchronize()

    def run(full):
        c = api.Context(0)
        try:
            c.set_graph(g["edges"], g["eweights"], n)
            c.set_embedding_device(X.data_ptr(), n, d, row_major=True)
            c.set_vertex_data(g["comm"], g["vweights"])
            res = c.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)
            out = dict(res=res, tr=c.last_trace, hi=c.last_diameter(),
                       pair=(c.get_stat("diameter_arg_i"), c.get_stat("diameter_arg_j")))
            if full:
                out["again"] = c.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)
                out["lm"] = c.landmarks_fetch()
            return out
        finally:
            c.close()

    _GRAPHS.clear()
    r1 = run(1)
    res, tr = r1["res"], r1["tr"]
    hi, path, pairs, tiles = r1["hi"]
    assert path == "pruned" and np.array_equal(res, r1["again"])
    dii, lemb, lcomm, ledges, lw, lweight, v2l = r1["lm"]
    N = len(dii)
    assert N == land and v2l.min() == 1 and v2l.max() == N
    comm = g["comm"][:, 0]
    first = np.zeros(N + 1, dtype=np.int64)
    first[v2l] = comm
    assert np.array_equal(first[v2l], comm) and np.array_equal(lcomm[:, 0], first[1:])  # landmarks nest in communities
    assert lw.sum() == m and lweight.sum() == 2 * m and np.array_equal(np.bincount(v2l, weights=g["vweights"])[1:], lweight)
    order = np.argsort(v2l, kind="stable")
    starts = np.concatenate([[0], np.cumsum(np.bincount(v2l)[1:])])
    for l in np.random.default_rng(1).integers(1, N + 1, 25):
        mem = order[starts[l - 1]:starts[l]]
        rows = X[torch.from_numpy(mem).to(dev)].cpu().numpy()
        w = g["vweights"][mem]
        cen = (rows * w[:, None]).sum(0) / w.sum()
        assert np.allclose(lemb[l - 1], cen, rtol=1e-12, atol=1e-14)
        assert dii[l - 1] == pytest.approx(np.sqrt(((rows - lemb[l - 1]) ** 2).sum() / w.sum()), rel=1e-12)
    assert np.all(np.isfinite(res)) and 0.25 <= res[0] <= 10 and 0 < res[1] <= math.log(2) and 0 <= res[5] <= 1
    best = int(np.nanargmin(tr["div"]))
    assert res[1] == tr["div"][best] and res[0] == 0.25 * (best + 1) and all(it >= 1 for it in tr["iters"])
    # the diameter: a 10^5-row sample that contains the arg-max pair, through the CPU branch and bound (dist()'s own bits)
    pi, pj = r1["pair"]
    assert 1 <= pi <= n and 1 <= pj <= n and pi != pj
    rows = np.unique(np.concatenate([np.random.default_rng(2).integers(0, n, 100_000), [pi - 1, pj - 1]]))
    Xs = np.asfortranarray(X[torch.from_numpy(rows).to(dev)].cpu().numpy())
    ref_hi, _, _, _ = exact_diameter(Xs, comm[rows])
    assert ref_hi == hi, (ref_hi, hi)
    del dii, lemb, lcomm, ledges, lw, lweight, r1
    # a fresh context: the same diameter, the same score, bit for bit
    r0 = run(0)
    assert r0["hi"][0] == hi and np.array_equal(r0["res"], res)


def test_config5_d512_twelve_thousand_landmarks(ctx):
    """configs[4] (ABCD 10M / 200M, d = 512, -l 12000) at the shape the test budget allows: n = 200 000 vertices, 4.2M edges
    asked for, 1500 communities, d = 512, 12 000 landmarks, i.e. the code paths of that configuration (the batched eigen-solver
    for 128 < d <= 512, the fp32-MFMA bound pass at K = 512, the 12 000-landmark sweep on the launch-per-iteration fit, vect_B
    by tiles) -- the full-size run is `bench.py --workload cfg5` (the 41 GB embedding is generated in HBM).  Round 5: PINNED
    against the oracle's run of exactly this graph (tests/golden/oracle_cfg5_200k.npz, make_oracle_fixture_fullsize.py
    cfg5_200k: 12 000 splits with LAPACK's syevr eigenvectors, 16 CPU-minutes) -- raw landmark ids, d_ii, weights, communities,
    checksums, iteration counts, the 7-vector and its traces -- plus: the exact diameter against the CPU branch and bound,
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
    # the oracle's run of this graph (src/landmarks.jl:160-209, :279-345; src/divergence.jl:139-256)
    fx = _fixture("cfg5_200k")
    assert g["n"] == int(fx["n"]) and g["m"] == int(fx["m"]) and crc(g["edges"]) == int(fx["edges_crc"]) and \
        crc(g["embedding"]) == int(fx["emb_crc"]), "the synthetic generator no longer reproduces the fixture's graph"
    assert hi == float(fx["hi"])
    lm = _check_landmarks(ctx, fx)
    _check_sweep(res, tr, fx, same_samples=False)
    smp = random_samples(np.random.default_rng(42), g["m"], g["n"], 10000)
    res_h = ctx.wgcl(lm[3], lm[4], lm[2], lm[1], lm[0], lm[5], g["vweights"], lm[6], None, None, None, False, auc_samples=10000,
                     directed=False, samples=smp, use_resident_original=True)
    _check_sweep(res_h, ctx.last_trace, fx, same_samples=True)
    try:  # bound pass in fp64, then brute force: the same diameter bits, the same score
        ctx.set_option("diameter_f32", 0)
        assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
        assert ctx.last_diameter()[0] == hi
        ctx.set_option("diameter", 1)
        assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
        assert ctx.last_diameter()[:2] == (hi, "brute")
    finally:
        ctx.set_option("diameter", 0)
        ctx.set_option("diameter_f32", 2)
    # the embedding handed over as a device pointer (row-major torch tensor), as bench.py --workload cfg5 does
    X = torch.from_numpy(np.ascontiguousarray(g["embedding"])).to("cuda:0")
    torch.cuda.synchronize()
    ctx.set_embedding_device(X.data_ptr(), n, d, row_major=True)
    del X
    assert np.array_equal(res, ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000))
