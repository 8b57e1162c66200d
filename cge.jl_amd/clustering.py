"""Host-side mirror of ``louvain_clust`` (reference: src/clustering.jl:14-68).

The reference shells out to three executables of ``louvain_jll`` through files in /tmp and leaves the LEVEL-1 communities
(``hierarchy -l 1``) in ``<edges file>.ecg``, one "node community" pair per line; ``parseargs`` calls it when the command
line has no ``-c`` (src/auxilary.jl:115-121).  Here the communities come from the device (``cge_louvain``,
csrc/kernels_louvain.hip) and the same file is written.  The reference visits the vertices in an unseeded random order,
so its own runs do not agree with each other; quality (modularity) is what the tests compare.
"""
from __future__ import annotations

import numpy as np


def _write_ecg(path, ids, comm):
    with open(path, "w") as f:
        f.write("\n".join(f"{int(i)} {int(c)}" for i, c in zip(ids, comm)))
        f.write("\n")


def _communities(edges1, weights, ctx):
    """edges1: (m, 2) 1-based int64; returns comm (n,) for vertices 1..n, 0-based consecutive labels."""
    from . import api

    ctx = ctx or api.default_context()
    n = int(edges1.max())
    ctx.set_graph(np.asfortranarray(edges1), np.ones(len(edges1)) if weights is None else weights, n)
    comm, n_comm, q, rounds = ctx.louvain()
    return comm, n_comm, q


def louvain_clust(*args, ctx=None):
    """``louvain_clust(v_min::Float64, edges::String)`` (src/clustering.jl:14-29) or
    ``louvain_clust(filename::String, edges::Array{Int,2}, weights::Array{Float64,1})`` (:42-68).  Writes
    ``<file>.ecg`` with one "vertex community" line per vertex, vertex ids in the edge file's own base (the reference
    drops the phantom vertex 0 of a 1-based file, :25-28)."""
    from . import api

    if len(args) == 2:
        v_min, path = args
        raw, _ = api.read_table(path, column_major=False)
        e = raw[:, :2].astype(np.int64)
        shift = 1 if float(v_min) == 0.0 else 0
        comm, _, _ = _communities(e + shift, None, ctx)
        n = len(comm)
        _write_ecg(path + ".ecg", np.arange(n) + (1 - shift), comm)
        return path + ".ecg"
    if len(args) == 3:
        filename, edges, weights = args
        comm, _, _ = _communities(np.asarray(edges, dtype=np.int64), np.asarray(weights, dtype=np.float64), ctx)
        _write_ecg(filename + ".ecg", np.arange(len(comm)) + 1, comm)  # the weighted form always drops row 0 (:67)
        return filename + ".ecg"
    raise TypeError("louvain_clust(v_min, edges_file) or louvain_clust(filename, edges, weights)")
