"""Two ranks on the one GPU of the test box (gloo between them): the N > 1 path of the library itself -- the forced
per-community phase of runsplit split over the ranks, edge shards, the sharded centroid pass and candidate tiles of the
diameter, the four all-reduces through the collective hook --
must reproduce the one-rank score.  (RCCL over xGMI needs several GPUs; the driver's scaling run covers that.  Two
processes cannot both keep a persistent grid resident on one GPU, so the ranks use the launch-per-iteration fit.)"""
import os
import socket
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cge.jl_amd import api

    c = api.Context(0)
    yield c
    c.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _graph():
    from cge.jl_amd import synth

    return synth.abcd_like(30000, 300000, 30, 16, seed=21)


def _rank(rank, world, port, q, mode, method="rss", shard_samples=1, fit_persistent=1, timeout_rank=-1, shard_ingest=0):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from cge.jl_amd import api
        from cge.jl_amd.dist import TorchCollectives

        g = _graph()
        ctx = api.Context(0)
        if shard_ingest:  # collectives first: the uploads themselves are split over the ranks
            coll = TorchCollectives(ctx, 600 * 600 * 2 + 1024, torch.device("cuda", 0))
            ctx.set_option("shard_ingest", 1)
            if shard_ingest == 2:  # a weighted list (dyadic weights: the sums stay exact whatever their grouping)
                g["eweights"] = 1.0 + (np.arange(g["m"]) % 4) * 0.25
            ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        else:
            ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
            coll = TorchCollectives(ctx, 600 * 600 * 2 + 1024, torch.device("cuda", 0))
        ctx.set_option("fit_persistent", fit_persistent)
        if rank == timeout_rank:  # this rank's persistent fits give up at once; the other rank's succeed
            ctx.set_option("fit_persistent_test_timeout", 1)
        ctx.set_option("shard_runsplit", mode)  # 2: every batch of the global phase is split too, whatever its size
        ctx.set_option("shard_samples", shard_samples)  # 2: the local-score tallies of every alpha are split over the ranks
        res = ctx.score(g["clusters"], 600, 2, method, seed=5, auc_samples=4000)
        hi = ctx.last_diameter()[0]
        batches = ctx.get_stat("landmark_batches")
        n0 = coll.n_calls
        v2l = ctx.landmarks_fetch()[6]
        q.put((rank, res.tolist(), hi, (n0, coll.n_calls - n0, batches, ctx.get_stat("fit_persistent_alphas"),
                                        ctx.get_stat("edges_resident"), ctx.get_stat("edges_total")),
               int(zlib.crc32(v2l.tobytes()))))
        ctx.close()
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e), None, None, None))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("mode", [1, 2, 0])
def test_two_ranks_on_one_gpu_reproduce_one_rank(ctx, mode):
    import torch.multiprocessing as mp

    g = _graph()
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=4000)
        hi_ref = ctx.last_diameter()[0]
        crc_ref = int(zlib.crc32(ctx.landmarks_fetch()[6].tobytes()))
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    for rank, res, hi, n_calls, crc in results:
        assert hi is not None, res  # a traceback otherwise
        assert crc == crc_ref  # v_to_l: the forced phase of runsplit split over the ranks gives the same landmark ids
        assert hi == hi_ref  # the exact diameter: a maximum over shards
        assert res[0] == ref[0] and res[4] == ref[4]
        assert np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
        # vect_C (sum), the centroid bounds (max), the diameter (max); + the forced phase's groups (gather) + one gather per
        # split batch of the global phase; the fetch adds the landmark-pair matrix (sum)
        during, fetch, batches = n_calls[:3]
        assert fetch == 2 and during >= 3 + (mode > 0)  # (fetch: the matrix + the count of its positive entries over the row blocks)
        if mode == 2:
            assert during > 4  # batches of the global phase went through the exchange
        if mode == 0:
            assert during == 3
    assert results[0][1] == results[1][1]  # both ranks hold the same bits


@pytest.mark.parametrize("method", ["rss2", "size", "diameter"])
def test_two_ranks_other_split_rules(ctx, method):
    """The same with every batch of runsplit split over the two ranks, for the other three split rules -- and with the
    local-score tallies of every alpha split over the ranks (S / W samples per rank, an all-reduce of the block
    tallies per alpha; unit weights, so the sums are exact whatever their grouping)."""
    import torch.multiprocessing as mp

    g = _graph()
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, method, seed=5, auc_samples=4000)
        crc_ref = int(zlib.crc32(ctx.landmarks_fetch()[6].tobytes()))
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, q, 2, method, 2)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    for rank, res, hi, n_calls, crc in results:
        assert hi is not None, res
        assert crc == crc_ref and np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
        assert n_calls[0] > 3 + 10  # ... plus one all-reduce of the tallies per alpha with a local score
    assert results[0][1] == results[1][1]


def test_two_ranks_fit_abandoned_on_one_rank_only(ctx):
    """with the tallies split over the ranks every alpha carries an all-reduce, and the verdict of an ENQUEUED
    persistent fit is rank-local.  Rank 0's persistent fits are made to give up at once (testing option), rank 1's succeed:
    the verdict travels with the all-reduced tallies, so both ranks redo the first alpha together with one launch per
    iteration, issue the same number of exchanges and end with the one-rank result (no hang, no mixed-up tallies)."""
    import torch.multiprocessing as mp

    g = _graph()
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=4000)
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, q, 1, "rss", 2, 2, 0)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    for rank, res, hi, n_calls, crc in results:
        assert hi is not None, res
        assert np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
    assert results[0][1] == results[1][1]
    assert results[0][3][0] == results[1][3][0]  # the same number of exchanges on both ranks
    assert results[0][3][3] == 0 and results[1][3][3] == 0  # no alpha was taken from a persistent fit: both ranks fell back together


@pytest.mark.parametrize("weighted", [0, 1])
def test_two_ranks_sharded_ingest(ctx, weighted):
    """Option shard_ingest (north star: "edge list and embedding rows shard across the GPUs"): every rank uploads HALF of the
    edge list and keeps only that (its scatter passes run over what it holds; the sampler's edge look-ups and non-edge checks
    are exchanged), and uploads half of the embedding's rows, which are all-gathered device to device.  The score, the
    diameter and the landmark ids are those of one rank holding everything."""
    import torch.multiprocessing as mp

    g = _graph()
    if weighted:
        g["eweights"] = 1.0 + (np.arange(g["m"]) % 4) * 0.25
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=4000)
        hi_ref = ctx.last_diameter()[0]
        lm_ref = ctx.landmarks_fetch()
        crc_ref = int(zlib.crc32(lm_ref[6].tobytes()))
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, q, 1, "rss", 1, 1, -1, 1 + weighted)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    held = 0
    for rank, res, hi, n_calls, crc in results:
        assert hi is not None, res  # a traceback otherwise
        assert crc == crc_ref and hi == hi_ref
        assert res[0] == ref[0] and res[4] == ref[4]
        assert np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
        assert n_calls[5] == g["m"] and n_calls[4] in (g["m"] // 2, g["m"] - g["m"] // 2)
        held += n_calls[4]
    assert held == g["m"]  # every edge is resident on exactly one rank
    assert results[0][1] == results[1][1]


# ---- option shard_rows: the embedding rows sharded by community (north star: "edge list and embedding rows shard across the GPUs") ----
def _graph_rows(case):
    """Inputs of the sharded-rows cases: the 30 000-vertex graph; `directed`; `weighted` (dyadic weights: sums stay exact whatever
    their grouping); `dups`: only 400 distinct embedding rows, spread over all communities (so equal rows live on BOTH ranks):
    `land` = 600 is clamped to the unique-row count (src/landmarks.jl:371-376) through the exact cross-rank comparison."""
    from cge.jl_amd import synth

    g = synth.abcd_like(30000, 300000, 30, 16, seed=21, directed=case.get("directed", False))
    if case.get("weighted"):
        g["eweights"] = 1.0 + (np.arange(g["m"]) % 4) * 0.25
        vw = np.zeros(g["n"])
        np.add.at(vw, g["edges"][:, 0] - 1, g["eweights"])
        np.add.at(vw, g["edges"][:, 1] - 1, g["eweights"])
        g["vweights"] = vw
    if case.get("dups"):
        g["embedding"] = np.asfortranarray(g["embedding"][np.arange(g["n"]) % 400])
    return g


def _rank_rows(rank, world, port, q, case):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from cge.jl_amd import api
        from cge.jl_amd.dist import TorchCollectives, community_owner

        g = _graph_rows(case)
        ctx = api.Context(0)
        coll = TorchCollectives(ctx, 600 * 600 * 2 + 1024, torch.device("cuda", 0))  # collectives first: the uploads are split
        ctx.set_option("shard_ingest", 0 if case.get("wgcl") else 1)  # (caller-drawn samples index the whole edge list)
        ctx.set_option("shard_rows", 1)
        ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        ctx.set_option("fit_persistent", 1)  # (two processes cannot both keep a persistent grid resident on one GPU)
        ctx.set_option("shard_samples", case.get("shard_samples", 1))
        directed = bool(case.get("directed", False))
        res = ctx.score(g["clusters"], 600, 2, case.get("method", "rss"), directed=directed, seed=5, auc_samples=4000)
        hi = ctx.last_diameter()[0]
        rows = (ctx.get_stat("rows_resident"), ctx.get_stat("rows_total"), ctx.get_stat("embedding_words_resident"))
        owner = community_owner(g["comm"][:, 0], world)
        mine = int(np.sum(owner[g["comm"][:, 0] - 1] == rank))
        lm = ctx.landmarks_fetch()
        extra = None
        if case.get("wgcl"):  # the reference's call shape, wGCL(landmark graph, v_to_l, ...), on the sharded rows: the landmark
            # index is rebuilt from the caller's v_to_l (this rank's members), the diameter and the sampled pairs' rows go
            # through the same exchanges
            dii, lemb, lcomm, ledges, lw, lweight, v2l = lm
            rng = np.random.default_rng(9)
            smp = (rng.integers(1, g["m"] + 1, size=(1, 3000)), rng.integers(1, g["n"] + 1, size=(1, 3000)),
                   rng.integers(1, g["n"] + 1, size=(1, 3000)))
            smp[2][smp[1] == smp[2]] = smp[2][smp[1] == smp[2]] % g["n"] + 1
            r2 = ctx.wgcl(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, None, None, None, False, auc_samples=3000,
                          samples=smp, use_resident_original=True)
            # ... and what sharded rows refuse, with a message instead of a hang: exact mode (every row on every rank), and
            # clusters that do not refine the community vector (here: two communities of different ranks merged) -- the
            # verdict is exchanged, so BOTH ranks leave by the same door
            errs = []
            for bad in ("exact", "span"):
                try:
                    if bad == "exact":
                        ctx.score(g["clusters"], -1, 2, "rss", seed=5, auc_samples=1000)
                    else:
                        a, b = int(np.flatnonzero(owner == 0)[0]), int(np.flatnonzero(owner == 1)[0])
                        cl = [c for k, c in enumerate(g["clusters"]) if k not in (a, b)] + [np.sort(np.concatenate([g["clusters"][a], g["clusters"][b]]))]
                        ctx.score(cl, 600, 2, "rss", seed=5, auc_samples=1000)
                    errs.append("no error")
                except api.CGEError as e:
                    errs.append((e.code, "shard_rows" in str(e)))
            extra = (r2.tolist(), ctx.last_diameter()[0], errs)
        q.put((rank, res.tolist(), hi, rows + (mine, ctx.truncated, ctx.get_stat("edges_resident"), extra),
               [int(zlib.crc32(np.ascontiguousarray(x).tobytes())) for x in lm]))
        ctx.close()
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e), None, None, None))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


ROW_CASES = [dict(method="rss"), dict(method="rss", wgcl=True), dict(method="rss2"), dict(method="size"), dict(method="diameter"),
             dict(method="rss", directed=True), dict(method="rss", weighted=True), dict(method="rss", shard_samples=2),
             dict(method="rss", dups=True), dict(method="diameter", directed=True, weighted=True)]


@pytest.mark.parametrize("case", ROW_CASES, ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()))
def test_two_ranks_sharded_rows(ctx, case):
    """rank r keeps the rows of ITS communities only (stat rows_resident ~ n / 2) and runs all their
    splits, forced and global; the heap is replicated from (status, size, value) words; v_to_l, the landmark tables, the
    diameter's bound matrix / seed row / candidate rows, the sampled pairs' rows and the row hashes are exchanged.  All four
    split rules, directed, weighted, the tallies split, and a unique-row clamp whose equal rows live on both ranks: the
    one-rank v_to_l, diameter bits, landmark tables and 7-vector, the same bits on both ranks."""
    import torch.multiprocessing as mp

    g = _graph_rows(case)
    directed = bool(case.get("directed", False))
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, case.get("method", "rss"), directed=directed, seed=5, auc_samples=4000)
        hi_ref = ctx.last_diameter()[0]
        lm_ref = ctx.landmarks_fetch()
        trunc_ref = ctx.truncated
        crc_ref = [int(zlib.crc32(np.ascontiguousarray(x).tobytes())) for x in lm_ref]
        extra_ref = None
        if case.get("wgcl"):
            dii, lemb, lcomm, ledges, lw, lweight, v2l = lm_ref
            rng = np.random.default_rng(9)
            smp = (rng.integers(1, g["m"] + 1, size=(1, 3000)), rng.integers(1, g["n"] + 1, size=(1, 3000)),
                   rng.integers(1, g["n"] + 1, size=(1, 3000)))
            smp[2][smp[1] == smp[2]] = smp[2][smp[1] == smp[2]] % g["n"] + 1
            r2 = ctx.wgcl(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, None, None, None, False, auc_samples=3000,
                          samples=smp, use_resident_original=True)
            extra_ref = (r2.tolist(), ctx.last_diameter()[0])
    finally:
        ctx.set_option("fit_persistent", 0)
    if case.get("dups"):
        assert trunc_ref and len(lm_ref[0]) == 400  # the clamp really applied
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_rows, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    held = 0
    for rank, res, hi, rows, crcs in results:
        assert hi is not None, res  # a traceback otherwise
        resident, total, words, mine, trunc, edges_res, extra = rows
        assert total == g["n"] and resident == mine and words == resident * 16  # only this rank's rows are in HBM ...
        assert 0.35 * total < resident < 0.65 * total  # ... about half of them
        assert case.get("wgcl") or edges_res < g["m"]  # (and its slice of the edge list)
        if case.get("wgcl"):
            assert extra[1] == extra_ref[1] and extra[0][0] == extra_ref[0][0] and extra[0][4] == extra_ref[0][4]
            assert np.allclose(extra[0], extra_ref[0], rtol=1e-12, atol=1e-14), (extra, extra_ref)
            assert extra[2] == [(-7, True), (-7, True)], extra[2]  # CGE_E_ARG, naming the option, on both ranks
        held += resident
        assert trunc == trunc_ref
        assert crcs[6] == crc_ref[6], "v_to_l differs from the one-rank run"
        assert hi == hi_ref  # the exact diameter, bit for bit
        # d_ii, centroids, communities, landmark edge list, weights, landmark weights: the one-rank bits (dyadic / unit weights)
        assert crcs == crc_ref
        assert res[0] == ref[0] and res[4] == ref[4]
        assert np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
    assert held == g["n"]  # every row is resident on exactly one rank
    assert results[0][1] == results[1][1]  # both ranks hold the same bits


# ---- round 5: the branches that only a reduce-scatter / all-gather reaches, the owner-only error, four ranks ----------------
def _rank_ext(rank, world, port, q, directed):
    """Sharded ingest + `wedges_reduce_scatter` through the hook's OWN all-gather / reduce-scatter (dist.TorchCollectives(ext=True):
    the reduce-scatter leaves NaN in every block a rank does not own)."""
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from cge.jl_amd import api, synth
        from cge.jl_amd.dist import TorchCollectives

        g = synth.abcd_like(30000, 300000, 30, 16, seed=21, directed=directed)
        ctx = api.Context(0)
        coll = TorchCollectives(ctx, 30000 * 16 + 1024, torch.device("cuda", 0), ext=True)  # (room for the embedding's all-gather)
        ctx.set_option("shard_ingest", 1)
        ctx.set_option("wedges_reduce_scatter", 1)
        ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        ag_ingest = coll.n_gather
        ctx.set_option("fit_persistent", 1)
        res = ctx.score(g["clusters"], 600, 2, "rss", directed=directed, seed=5, auc_samples=4000)
        hi = ctx.last_diameter()[0]
        rs_score, ag_score = coll.n_reduce_scatter, coll.n_gather
        lm = ctx.landmarks_fetch()  # collective with the matrix in row blocks: the blocks are all-gathered
        q.put((rank, res.tolist(), hi, (ag_ingest, rs_score, ag_score, coll.n_reduce_scatter, coll.n_gather),
               [int(zlib.crc32(np.ascontiguousarray(x).tobytes())) for x in lm]))
        ctx.close()
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e), None, None, None))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("directed", [False, True])
def test_two_ranks_reduce_scatter_and_allgather_through_the_hook(ctx, directed):
    """VERDICT r4 item 4(a) / ADVICE r4: what runs only behind a reduce-scatter of the N x N landmark-pair matrix -- the count of
    its positive entries over row blocks, the directed score's degrees from row blocks (k_wedge_degrees_block), the all-gather
    of the blocks in cge_landmarks_fetch -- and the all-gather of the sharded ingest, executed through the hook's own
    reduce-scatter (which poisons every foreign block with NaN) and all-gather: the one-rank result, tables and edge list."""
    import torch.multiprocessing as mp
    from cge.jl_amd import synth

    g = synth.abcd_like(30000, 300000, 30, 16, seed=21, directed=directed)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, "rss", directed=directed, seed=5, auc_samples=4000)
        hi_ref = ctx.last_diameter()[0]
        crc_ref = [int(zlib.crc32(np.ascontiguousarray(x).tobytes())) for x in ctx.landmarks_fetch()]
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_ext, args=(r, 2, port, q, directed)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    for rank, res, hi, calls, crcs in results:
        assert hi is not None, res  # a traceback otherwise
        ag_ingest, rs_score, ag_score, rs_all, ag_all = calls
        assert ag_ingest == 1  # the embedding's rows: one all-gather
        assert rs_all >= 1 and ag_all == ag_score + 1  # the matrix went out by row blocks; the fetch gathered them
        assert (rs_score >= 1) == directed or not directed  # (the directed score needs the matrix itself: degrees from row blocks)
        assert hi == hi_ref and crcs == crc_ref
        assert res[0] == ref[0] and res[4] == ref[4] and np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
    assert results[0][1] == results[1][1]


def _rank_homogeneous(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from cge.jl_amd import api
        from cge.jl_amd.dist import TorchCollectives, community_owner

        g = _graph_homogeneous()
        ctx = api.Context(0)
        TorchCollectives(ctx, 600 * 600 * 2 + 1024, torch.device("cuda", 0))
        ctx.set_option("shard_ingest", 1)
        ctx.set_option("shard_rows", 1)
        ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        ctx.set_option("fit_persistent", 1)
        owner = int(community_owner(g["comm"][:, 0], world)[g["hom"] - 1])
        try:
            ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=1000)
            q.put((rank, "no error", owner))
        except api.CGEError as e:
            q.put((rank, (e.code, str(e)), owner))
        ctx.close()
    except Exception as e:
        import traceback

        q.put((rank, traceback.format_exc() + repr(e), None))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


def _graph_homogeneous():
    """The 30 000-vertex graph with ONE community whose rows are all equal (small integers: with integer degrees as weights the
    weighted mean is exact): the rss rule must refuse to split it (src/landmarks.jl:165-167, "Trying to split homogenous cluster")."""
    g = _graph()
    comm = g["comm"][:, 0]
    hom = int(np.argmin(np.bincount(comm)[1:]) + 1)  # the smallest community
    emb = np.array(g["embedding"], order="F")
    emb[comm == hom] = np.arange(1.0, emb.shape[1] + 1.0)  # integer coordinates: the weighted mean is exact, every z is 0 whatever the order
    g["embedding"] = np.asfortranarray(emb)
    g["hom"] = hom
    return g


def test_two_ranks_owner_only_error_reaches_both_ranks(ctx):
    """with the rows sharded by community the reference's own errors are raised by the OWNER of the community
    alone, between two collectives -- the other rank used to wait in the next exchange for ever.  The verdict now travels with
    the exchange: both ranks stop with the owner's code (CGE_E_HOMOGENEOUS), the owner with the reference's message."""
    import torch.multiprocessing as mp
    from cge.jl_amd import api

    g = _graph_homogeneous()
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    with pytest.raises(api.CGEError) as ei:  # one rank: the reference's error
        ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=1000)
    assert ei.value.code == -2 and "homogenous" in str(ei.value)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_homogeneous, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)  # (a hang would end here)
    for p in procs:
        p.join(120)
    for rank, err, owner in results:
        assert owner is not None and err != "no error", err
        assert err[0] == -2, err
        assert "homogenous" in err[1] and (f"rank {owner} failed" in err[1]) == (rank != owner), err  # (the code maps to the reference's message)


def test_four_ranks_sharded_rows(ctx):
    """The sharded-rows score on FOUR ranks (four processes on the test box's one GPU, gloo between them): the ownership of 30
    communities over four owners, row blocks of 150 landmarks, four-way edge and tile shards -- the one-rank bits."""
    import torch.multiprocessing as mp

    case = dict(method="rss")
    g = _graph_rows(case)
    ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
    try:
        ctx.set_option("fit_persistent", 1)
        ref = ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=4000)
        hi_ref = ctx.last_diameter()[0]
        crc_ref = [int(zlib.crc32(np.ascontiguousarray(x).tobytes())) for x in ctx.landmarks_fetch()]
    finally:
        ctx.set_option("fit_persistent", 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_rows, args=(r, 4, port, q, case)) for r in range(4)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(120)
    held = 0
    for rank, res, hi, rows, crcs in results:
        assert hi is not None, res
        resident, total, words, mine = rows[:4]
        assert total == g["n"] and resident == mine and 0.15 * total < resident < 0.35 * total  # about a quarter each
        held += resident
        assert hi == hi_ref and crcs == crc_ref
        assert res[0] == ref[0] and res[4] == ref[4] and np.allclose(res, ref, rtol=1e-12, atol=1e-14), (rank, res, ref)
    assert held == g["n"] and all(r[1] == results[0][1] for r in results)
