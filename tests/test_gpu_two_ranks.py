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


def _rank(rank, world, port, q):
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
        ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
        coll = TorchCollectives(ctx, 600 * 600 * 2 + 1024, torch.device("cuda", 0))
        ctx.set_option("fit_persistent", 1)
        res = ctx.score(g["clusters"], 600, 2, "rss", seed=5, auc_samples=4000)
        hi = ctx.last_diameter()[0]
        v2l = ctx.landmarks_fetch()[6]
        q.put((rank, res.tolist(), hi, coll.n_calls, int(zlib.crc32(v2l.tobytes()))))
        ctx.close()
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e), None, None, None))
    finally:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_one_rank(ctx):
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
    procs = [mpc.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
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
        # the forced phase's groups (gather), vect_C (sum), the centroid bounds (max), the diameter (max); the fetch above
        # adds the landmark-pair matrix (sum)
        assert n_calls == 5
    assert results[0][1] == results[1][1]  # both ranks hold the same bits
