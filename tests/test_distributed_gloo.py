"""world_size = 2 over gloo on the CPU: the sharding rules and the collective hook of the N > 1 path.
The GPU kernels are not involved (no GPU here); each rank computes its shard with numpy (the oracle's
arithmetic for unit weights is exact integer counting) and the reductions must reproduce the whole."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cge.jl_amd import dist as cd
        from cge.jl_amd import synth

        g = synth.abcd_like(3000, 24000, 6, 8, seed=3)
        m, n, C = g["m"], g["n"], g["C"]
        comm = g["comm"][:, 0]
        rng = np.random.default_rng(0)
        v2l = np.zeros(n, dtype=np.int64)
        for c in range(1, C + 1):
            idx = np.flatnonzero(comm == c)
            v2l[idx] = rng.integers(0, 4, size=len(idx)) + 4 * (c - 1)
        N = 4 * C

        def scatter(e0, e1):  # src/landmarks.jl:448-451 + src/divergence.jl:59-63 on rows [e0, e1)
            a, b = v2l[g["edges"][e0:e1, 0] - 1], v2l[g["edges"][e0:e1, 1] - 1]
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            wed = np.zeros((N, N))
            np.add.at(wed, (lo, hi), g["eweights"][e0:e1])
            ca, cb = comm[g["edges"][e0:e1, 0] - 1] - 1, comm[g["edges"][e0:e1, 1] - 1] - 1
            vc = np.zeros((C, C))
            np.add.at(vc, (np.minimum(ca, cb), np.maximum(ca, cb)), g["eweights"][e0:e1])
            return wed, vc

        # 1. edge shards: contiguous, disjoint, covering; partial scatters all-reduce to the whole
        e0, e1 = cd.edge_shard(m, rank, world)
        bounds = [cd.edge_shard(m, r, world) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == m and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        wed, vc = scatter(e0, e1)
        wed_all = cd.allreduce_numpy(wed.copy(), "sum")
        vc_all = cd.allreduce_numpy(vc.copy(), "sum")
        full_w, full_c = scatter(0, m)
        assert np.array_equal(wed_all, full_w) and np.array_equal(vc_all, full_c)
        assert wed_all.sum() == g["eweights"].sum()
        # 2. diameter shards: every pair tile / candidate tile is owned by exactly one rank; max of shard maxima
        nS = 7
        owned = [set(cd.diameter_shard(nS, r, world)) for r in range(world)]
        assert set().union(*owned) == set(range(nS)) and sum(map(len, owned)) == nS
        tiles = [set(cd.candidate_tile_shard(1001, r, world)) for r in range(world)]
        assert set().union(*tiles) == set(range(1001)) and sum(map(len, tiles)) == 1001
        # 2b. the centroid pass: contiguous vertex-tile ranges that cover all tiles; max over vertices of the distance to
        #     every reference point = max over ranks of the per-shard maxima
        nT = (n + 127) // 128
        spans = [cd.centroid_tile_shard(nT, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == nT and all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        refs = g["embedding"][:: max(1, n // 7)][:7]
        t0, t1 = spans[rank]
        part = g["embedding"][128 * t0: min(n, 128 * t1)]
        q_loc = np.array([((part - r_) ** 2).sum(1).max() if len(part) else 0.0 for r_ in refs])
        q_all = cd.allreduce_numpy(q_loc.copy(), "max")
        q_ref = np.array([((g["embedding"] - r_) ** 2).sum(1).max() for r_ in refs])
        assert np.array_equal(q_all, q_ref)
        X = g["embedding"]
        rows = np.arange(n)
        mine = rows[rows % world == rank]
        local = 0.0
        for i in mine[:: max(1, len(mine) // 200)]:  # a sample of this rank's rows against all rows
            local = max(local, np.sqrt(((X[i] - X) ** 2).sum(1)).max())
        glob = cd.allreduce_numpy(np.array([local]), "max")[0]
        assert glob >= local and glob == max(dist_gather(local, world))
        # 3. the hook itself (what the C library calls): offset arithmetic + op mapping on the exchange buffer
        coll = cd.TorchCollectives(None, 256, "cpu")
        tri = world * (world + 1) // 2
        coll.buf[:] = 0
        coll.buf[8:12] = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64) * (rank + 1)
        assert coll._hook(None, coll.base + 8 * 8, 4, 0) == 0
        assert coll.buf[8:12].tolist() == [1.0 * tri, 2.0 * tri, 3.0 * tri, 4.0 * tri] and coll.buf[12].item() == 0.0
        coll.buf[20] = float(10 + rank)
        assert coll._hook(None, coll.base + 20 * 8, 1, 1) == 0
        assert coll.buf[20].item() == 10.0 + world - 1 and coll.n_calls == 2 and coll.bytes == 40
        # op 2: the words as int64 -- disjoint shards in zero-filled buffers are gathered bit for bit (-0.0, NaN payloads, packed ints)
        coll.buf[30:30 + 2 * world] = 0
        mine = torch.tensor([-0.0, float("nan")], dtype=torch.float64) if rank == 0 else \
            torch.tensor([123456789 + rank], dtype=torch.int64).view(torch.float64).repeat(2)
        coll.buf[30 + 2 * rank: 32 + 2 * rank] = mine
        assert coll._hook(None, coll.base + 30 * 8, 2 * world, 2) == 0
        got = coll.buf[30:30 + 2 * world].view(torch.int64).tolist()
        assert got[0] == -(2 ** 63) and got[1] == torch.tensor([float("nan")], dtype=torch.float64).view(torch.int64).item()
        assert all(got[2 * r] == got[2 * r + 1] == 123456789 + r for r in range(1, world))
        # 3b. the optional ops of the hook (include/cge_hip.h: cge_collectives_ext), in place on the exchange buffer: the
        #     all-gather moves every rank's block bit for bit; after the reduce-scatter a rank may rely on ITS block only (the
        #     hook poisons the others, as the contract leaves them unspecified)
        w = 3
        blk = coll.buf[64:64 + w * world]
        blk[:] = -1.0
        blk[w * rank: w * (rank + 1)] = torch.tensor([rank + 0.5, -0.0, float(rank)], dtype=torch.float64)
        assert coll._hook_allgather(None, coll.base + 64 * 8, w) == 0
        for r in range(world):
            assert blk[w * r].item() == r + 0.5 and blk[w * r + 2].item() == float(r)
            assert blk[w * r + 1: w * r + 2].view(torch.int64).item() == -(2 ** 63)  # -0.0 kept its sign: words, not sums
        blk[:] = torch.arange(w * world, dtype=torch.float64) * (rank + 1)
        assert coll._hook_reduce_scatter(None, coll.base + 64 * 8, w) == 0
        assert blk[w * rank: w * (rank + 1)].tolist() == [float(tri * (w * rank + k)) for k in range(w)]
        others = torch.cat([blk[: w * rank], blk[w * (rank + 1):]])
        assert bool(torch.isnan(others).all()) and coll.n_gather == 1 and coll.n_reduce_scatter == 1
        # 4. option shard_rows: the ownership rule (rows sharded BY COMMUNITY) and the exchanges built on it.  Every rank derives
        #    the same owners from the replicated community vector; every vertex has exactly one owner; the loads are balanced;
        #    what a rank computes for ITS rows, written at their global ids into a zero-filled vector, is completed by the
        #    integer all-reduce (op 2) -- v_to_l, and per-landmark tables whose rows live on one rank each.
        owner = cd.community_owner(comm, world)
        assert np.array_equal(cd.allreduce_numpy(owner.copy(), "sum"), world * owner)  # the same on every rank
        mine_mask = owner[comm - 1] == rank
        assert np.array_equal(cd.allreduce_numpy(mine_mask.astype(np.int64), "sum"), np.ones(n, dtype=np.int64))
        loads = cd.allreduce_numpy(np.eye(world, dtype=np.int64)[rank] * int(mine_mask.sum()), "sum")
        sizes = np.bincount(comm - 1)
        assert loads.sum() == n and loads.max() - loads.min() <= sizes.max()  # greedy longest-first: within one community
        order = np.argsort(-sizes, kind="stable")
        assert owner[order[0]] == 0 and (world < 2 or owner[order[1]] == 1)  # largest first, ties to the lower rank
        part = np.zeros(n, dtype=np.int64)
        part[mine_mask] = v2l[mine_mask] + 1  # landmark + 1 of the rows this rank holds
        assert np.array_equal(cd.allreduce_numpy(part, "sum") - 1, v2l)
        lm_owner = owner[(np.arange(N) // 4)]  # a landmark lives where its community lives (v2l above: 4 landmarks per community)
        cen = np.zeros((N, 8))
        for l in np.flatnonzero(lm_owner == rank):
            cen[l] = X[v2l == l].sum(0) if np.any(v2l == l) else 0.0
        cen_ref = np.stack([X[v2l == l].sum(0) if np.any(v2l == l) else np.zeros(8) for l in range(N)])
        got = cd.allreduce_numpy(cen.view(np.int64).copy(), "sum").view(np.float64)
        assert np.array_equal(got, cen_ref)  # a gather, bit for bit
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def dist_gather(value, world):
    t = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(t, torch.tensor([value], dtype=torch.float64))
    return [float(x.item()) for x in t]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharding_and_collective_hook(world):
    """The sharding rules and the hook at the world sizes a node offers (2, 4, 8 ranks): edge / tile / community ownership
    partitions, every all-reduce op, the all-gather and reduce-scatter ops of the hook."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results
