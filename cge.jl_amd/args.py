"""Host-side mirror of ``parseargs()`` (reference: src/auxilary.jl:61-247).

Same flags, same defaults, same 14-tuple (src/auxilary.jl:220), 1-based ids.  The three input files are read by the
library's parallel text reader (csrc/textio.cpp) -- `readdlm` of a multi-GB embedding would
otherwise take longer than the scoring pass it feeds.  Parsing is still never part of the timed region.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

# The reference passes the split rule as a Julia function (src/auxilary.jl:64-67); the C-ABI takes
# an enum.  These sentinels keep call sites looking like the reference's (`CGE.split_cluster_rss2`).


class _SplitRule:
    def __init__(self, name: str, code: int):
        self.name, self.code = name, code

    def __repr__(self):
        return f"split_cluster_{self.name}"


split_cluster_rss = _SplitRule("rss", 0)
split_cluster_rss2 = _SplitRule("rss2", 1)
split_cluster_size = _SplitRule("size", 2)
split_cluster_diameter = _SplitRule("diameter", 3)
METHODS = {
    "rss": split_cluster_rss,
    "rss2": split_cluster_rss2,
    "size": split_cluster_size,
    "diameter": split_cluster_diameter,
}

USAGE = (
    "\n\nUsage:\n"
    "\tpython cge_cli.py -g edgelist -e embedding [-c communities] [--seed seed] [--samples-local samples] [-v] [-d] "
    "[--split-global] [-l [landmarks]] [-f [forced]] [--force-exact] [-m method]\n"
)


def _readdlm(path):
    """(table, header_skipped): every numeric field of the file; a node2vec "n d" header line is skipped."""
    from . import api

    try:
        return api.read_table(path, column_major=False)
    except api.CGEError as e:
        raise AssertionError(str(e)) from e


def _flag_value(argv, flag):
    if flag in argv:
        i = argv.index(flag)
        return argv[i + 1] if i + 1 < len(argv) else None
    return None


class ParseError(Exception):
    pass


def parseargs(argv=None, exit_on_error=True):
    """Returns (edges, eweights, vweight, comm, clusters, embedding, verbose, landmarks, forced,
    method, directed, split, seed, samples) exactly as src/auxilary.jl:220.

    edges: int64 (m,2) Fortran-ordered (two contiguous columns, like Julia's Matrix{Int});
    comm: int64 (n,1); embedding: float64 (n,d) Fortran-ordered; clusters: list of 1-based int64
    arrays (empty dict when no landmarks are requested, as in the reference :173,:199-208)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    try:
        verbose = "-v" in argv
        directed = "-d" in argv
        split = "--split-global" in argv
        if "-g" not in argv:
            raise AssertionError("Edgelist file is required")
        fn_edges = _flag_value(argv, "-g")
        if not os.path.isfile(fn_edges):
            raise AssertionError(f"{fn_edges} is not a file")
        raw, _ = _readdlm(fn_edges)
        rows, no_cols = raw.shape
        if no_cols not in (2, 3):
            raise AssertionError("Expected 2 or 3 columns in edgelist file")
        v_min = raw[:, :2].min()
        if v_min not in (0, 1):
            raise AssertionError("Vertices should be either 0-based or 1-based")
        if v_min == 0:
            raw[:, :2] += 1.0
        no_vertices = int(raw[:, :2].max())
        eweights = np.ones(rows) if no_cols == 2 else np.ascontiguousarray(raw[:, 2])
        edges = np.asfortranarray(raw[:, :2].astype(np.int64))
        vweight = np.zeros(no_vertices)
        # :107-110: `vweight[u] += w; vweight[v] += w` edge by edge -- the interleaved index list keeps that order of
        # additions (np.add.at is sequential), so non-dyadic weights round as in the reference
        np.add.at(vweight, np.ascontiguousarray(edges).ravel() - 1, np.repeat(eweights, 2))

        if "-c" in argv:
            fn_comm = _flag_value(argv, "-c")
        else:  # :115-121: no -c => Louvain communities (level 1) of the graph itself, written to <edgelist>.ecg
            from .clustering import louvain_clust

            if no_cols == 2:
                louvain_clust(float(v_min), fn_edges)
            else:
                louvain_clust(fn_edges, edges, eweights)
            fn_comm = fn_edges + ".ecg"
        comm_raw, _ = _readdlm(fn_comm)
        if not np.all(comm_raw == np.floor(comm_raw)):
            raise AssertionError("Communities file must hold integers")
        comm = comm_raw.astype(np.int64)
        comm_rows, ccols = comm.shape
        if comm_rows != no_vertices:
            raise AssertionError(f"No. communities ({comm_rows}) differ from no. nodes ({no_vertices})")
        if ccols not in (1, 2):
            raise AssertionError(f"Expected 1 or 2 columns in communities file, but encountered {ccols}.")
        if ccols == 2:
            comm = comm[np.argsort(comm[:, 0], kind="stable"), 1].reshape(-1, 1)
        c_min = comm.min()
        if c_min not in (0, 1):
            raise AssertionError(f"Communities should be either 0-based or 1-based, but are {c_min} based.")
        if c_min == 0:
            comm = comm + 1
        comm = np.asfortranarray(comm.astype(np.int64))

        if "-e" not in argv:
            raise AssertionError("Embedding file is required")
        fn_embed = _flag_value(argv, "-e")
        if not os.path.isfile(fn_embed):
            raise AssertionError(f"{fn_embed} is not a file")
        embedding, _ = _readdlm(fn_embed)  # a node2vec header line (:151-156) is skipped by the reader
        if embedding.shape[0] != no_vertices:
            raise AssertionError("No. rows in embedding and no. vertices in a graph differ.")
        first = embedding[:, 0]
        if np.all(first == np.floor(first)):  # convert.(Int, ...) succeeds (:161-167)
            embedding = embedding[np.argsort(first.astype(np.int64), kind="stable"), 1:]
        embedding = np.asfortranarray(embedding)

        landmarks = -1
        if "-l" in argv:
            try:
                landmarks = int(_flag_value(argv, "-l"))
            except (TypeError, ValueError):
                landmarks = int(round(4 * math.sqrt(no_vertices)))
                print(f"[ Info: Using {landmarks} landmarks", file=sys.stderr)
        if "-f" in argv:
            forced = int(_flag_value(argv, "-f"))
            landmarks = 1 if landmarks == -1 else landmarks
        else:
            forced = 4
        if no_vertices >= 10000 and "--force-exact" not in argv and landmarks == -1:
            landmarks = max(int(round(4 * math.sqrt(no_vertices))), 4 * int(comm.max()))
            print(
                f"[ Info: Number of vertices is equal or higher than 10 000. Automatically switching to approximate "
                f"algortihm with {landmarks} landmarks. If you want to force exact algorithm use --force-exact flag.",
                file=sys.stderr,
            )
        clusters = {}
        if landmarks != -1:
            c = comm[:, 0]
            order = np.argsort(c, kind="stable")
            bounds = np.flatnonzero(np.diff(c[order])) + 1
            clusters = [g.astype(np.int64) + 1 for g in np.split(order, bounds)]
        seed = int(_flag_value(argv, "--seed")) if "--seed" in argv else -1
        samples = int(_flag_value(argv, "--samples-local")) if "--samples-local" in argv else 10000
        method_str = _flag_value(argv, "-m").strip().lower() if "-m" in argv else "rss"
        method = METHODS[method_str]
        return (edges, eweights, vweight, comm, clusters, embedding, verbose, landmarks, forced, method,
                directed, split, seed, samples)
    except Exception as e:  # src/auxilary.jl:221-246: message + usage, exit(1)
        if not exit_on_error:
            raise ParseError(str(e)) from e
        print(f"{type(e).__name__}: {e}", file=sys.stderr)
        print(USAGE)
        sys.exit(1)
