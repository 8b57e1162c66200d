"""One process per GPU: sharding + the collective hooks of the C-ABI, over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

What is sharded:
  * runsplit: in the forced per-community phase and in the big batches of the global phase every rank cuts its share of
    the groups; one all-reduce per phase / batch (op 2: integer sum of the words of zero-filled buffers, i.e. an exact
    gather) hands every rank all results, and all ranks continue from identical state;
  * the per-edge scatter: rank r takes the edge rows [m*r/W, m*(r+1)/W); one all-reduce(sum) of the
    C x C cluster-pair vector vect_C (and of the N x N landmark-pair matrix when `landmarks_fetch` asks for it);
  * the point-set diameter: rank r takes its share of the vertex tiles of the centroid pass (one all-reduce(max) of
    the N x C bound matrix) and the candidate tiles t = r (mod W) / super-block rows SI = r (mod W) of the exact
    evaluation (one all-reduce(max) of a scalar);
and what is replicated: the heap of runsplit and its small batches, and the alpha sweep (sequentially
dependent iterations of a register-resident persistent fit that already uses every CU).  No other collective is issued.

Option "shard_rows" (set the collectives first, upload the communities before the embedding): the EMBEDDING ROWS are sharded by
community (community_owner below); a rank keeps and splits the rows of its own communities only; see include/cge_hip.h for
what crosses ranks.

Option "shard_ingest" (set the collectives first, then upload): `set_graph` keeps only the rows edge_shard(m, rank, world)
of the edge list on a rank (the scatter passes then run over what a rank holds; the sampler's edge look-ups and non-edge
checks are all-reduced), and `set_embedding` uploads n / world rows per rank and all-gathers them over xGMI
(ncclAllGather in the library; through this hook: op 2 on zero-filled pieces).
"""
from __future__ import annotations

import numpy as np


def edge_shard(m: int, rank: int, world: int):
    """Rows of the edge list owned by `rank` -- the same formula as capi.cpp:landmarks_run_impl."""
    return m * rank // world, m * (rank + 1) // world


def diameter_shard(n_super_rows: int, rank: int, world: int):
    """Super-block rows owned by `rank` in the brute-force kernel (kernels_dist.hip: SI % nparts == part)."""
    return [si for si in range(n_super_rows) if si % world == rank]


def centroid_tile_shard(n_vertex_tiles: int, rank: int, world: int):
    """Vertex tiles (128 rows each) of the diameter's centroid pass owned by `rank`
    (kernels_dist.hip: k_pcent, I0 = T*part/nparts, I1 = T*(part+1)/nparts); the per-landmark maxima are then
    combined by an all-reduce(max)."""
    return n_vertex_tiles * rank // world, n_vertex_tiles * (rank + 1) // world


def candidate_tile_shard(n_tiles: int, rank: int, world: int):
    """Candidate tiles owned by `rank` in the pruned diameter (diameter_host.cpp: global_tile % nparts == part)."""
    return list(range(rank, n_tiles, world))


def community_owner(comm, world: int):
    """Option shard_rows: owner rank of every community (index = community id, 1-based ids -> entry id - 1 ... here: returns
    an array indexed by the 0-based community) -- the same rule as capi.cpp:rows_assign_ownership: communities by decreasing
    size (ties: lower id first), each to the rank with the fewest rows so far (ties: lower rank).  `comm`: 1-based ids."""
    comm = np.asarray(comm).reshape(-1)
    size = np.bincount(comm - 1)
    order = np.argsort(-size, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(size), dtype=np.int64)
    for q in order:
        r = int(np.argmin(load))
        owner[q] = r
        load[r] += size[q]
    return owner


class TorchCollectives:
    """Implements cge_collectives.allreduce_f64 on a torch tensor that doubles as the library's
    exchange buffer (cge_set_exchange_buffer)."""

    def __init__(self, ctx, capacity_doubles: int, device, ext: bool = False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.buf = torch.zeros(int(capacity_doubles), dtype=torch.float64, device=device)
        self.base = self.buf.data_ptr()
        self.n_calls = 0
        self.bytes = 0
        self.n_gather = 0
        self.n_reduce_scatter = 0
        if ctx is not None:  # (None: the hook alone, as the CPU tests drive it)
            ctx._check(ctx.L.cge_set_exchange_buffer(ctx.h, _vp(self.base), _i64(self.buf.numel())))
            ctx.set_collectives(self._hook, self.rank, self.world)
            if ext:  # the optional all-gather / reduce-scatter of the hook (include/cge_hip.h: cge_collectives_ext)
                ctx.set_collectives_ext(self._hook_allgather, self._hook_reduce_scatter)

    def _hook(self, user, ptr, count, op):
        try:
            off = (int(ptr) - self.base) // 8
            t = self.buf[off: off + int(count)]
            rop = self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM
            if op == 2:  # the 8-byte words as integers: an exact gather of disjoint shards into zero-filled buffers
                t = t.view(self.torch.int64)
            if t.is_cuda and self.dist.get_backend() == "gloo":  # rehearsal on one GPU: stage through the host
                h = t.cpu()
                self.dist.all_reduce(h, op=rop)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=rop)
            if t.is_cuda:
                self.torch.cuda.synchronize(t.device)
            self.n_calls += 1
            self.bytes += int(count) * 8
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[cge.dist] allreduce hook failed: {e!r}", flush=True)
            return 1


    def _view(self, ptr, count):
        off = (int(ptr) - self.base) // 8
        return self.buf[off: off + int(count)]

    def _hook_allgather(self, user, ptr, words):
        """In place: rank r's block [r * words, (r + 1) * words) to every rank (8-byte words, moved as int64)."""
        try:
            t = self._view(ptr, int(words) * self.world).view(self.torch.int64)
            mine = t[self.rank * int(words): (self.rank + 1) * int(words)].clone()
            stage = t.is_cuda and self.dist.get_backend() == "gloo"  # rehearsal on one GPU: through the host
            if stage:
                mine = mine.cpu()
            parts = [self.torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine)
            t.copy_(self.torch.cat(parts).to(t.device))
            if t.is_cuda:
                self.torch.cuda.synchronize(t.device)
            self.n_gather += 1
            self.bytes += int(words) * 8 * self.world
            return 0
        except Exception as e:
            print(f"[cge.dist] all-gather hook failed: {e!r}", flush=True)
            return 1

    def _hook_reduce_scatter(self, user, ptr, words):
        """In place: afterwards block `rank` holds the sums of that block over the ranks.  The other blocks are UNSPECIFIED by the
        contract: they are filled with NaN here, so that a consumer that reads beyond its own block cannot pass a test."""
        try:
            w = int(words)
            t = self._view(ptr, w * self.world)
            stage = t.is_cuda and self.dist.get_backend() == "gloo"
            h = t.cpu() if stage else t
            blocks = [h[r * w: (r + 1) * w].clone() for r in range(self.world)]
            out = self.torch.empty_like(blocks[0])
            try:
                self.dist.reduce_scatter(out, blocks)
            except Exception:  # gloo has no reduce-scatter: all-reduce, keep the own block
                self.dist.all_reduce(h)
                out = h[self.rank * w: (self.rank + 1) * w].clone()
            t.fill_(float("nan"))
            t[self.rank * w: (self.rank + 1) * w].copy_(out.to(t.device))
            if t.is_cuda:
                self.torch.cuda.synchronize(t.device)
            self.n_reduce_scatter += 1
            self.bytes += w * 8 * self.world
            return 0
        except Exception as e:
            print(f"[cge.dist] reduce-scatter hook failed: {e!r}", flush=True)
            return 1


def _vp(x):
    import ctypes as C

    return C.c_void_p(x)


def _i64(x):
    import ctypes as C

    return C.c_int64(x)


def allreduce_numpy(arr: np.ndarray, op: str = "sum"):
    """Host-side reduction used by the gloo tests of the sharding logic."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(arr))
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.numpy()
