#!/usr/bin/env python3
"""cge_cli.py -- the reference's command-line script (example/CGE_CLI.jl:1-25) on the MI355X library.

Same flags (src/auxilary.jl:61-219, README.md:65-83), same call order, same output channels: the result vector is the
ONLY thing on stdout, printed the way Julia's `println(::Vector{Float64})` prints it (CGE_CLI.jl:25); one "." per
evaluated alpha and a newline go to stderr (src/divergence.jl:140,255).

    python cge_cli.py -g graph.edgelist -c graph.ecg -e graph.embedding -l 200 --seed 42
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def julia_float(x: float) -> str:
    """Float64 as Julia's `show` prints it: shortest round-trip digits, `1.0e-5` style exponents, Inf / NaN."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    r = repr(float(x))
    if "e" in r:
        mant, exp = r.split("e")
        if "." not in mant:
            mant += ".0"
        return f"{mant}e{int(exp)}"
    return r


def julia_vector(v) -> str:
    return "[" + ", ".join(julia_float(float(x)) for x in v) + "]"


def main(argv=None):
    import cge.jl_amd as CGE

    (edges, weights, vweights, comm, clusters, embed, verbose, land, forced, method, directed, split, seed,
     samples) = CGE.parseargs(argv)                                                  # CGE_CLI.jl:3
    distances = np.zeros(len(vweights))                                              # :4
    init_edges = np.zeros((0, 0), dtype=np.int64)                                    # :5
    init_vweights = np.zeros(0)                                                      # :6
    init_eweights = np.zeros(0)                                                      # :7
    init_embed = np.zeros((0, 0))                                                    # :8
    v_to_l = np.zeros(0, dtype=np.int64)                                             # :9
    if land != -1:                                                                   # :10
        init_edges, init_vweights, init_eweights, init_embed = edges.copy(), vweights.copy(), weights.copy(), embed.copy()
        distances, embed, comm, edges, weights, vweights, v_to_l = CGE.landmarks(
            edges, weights, vweights, clusters, comm, embed, verbose, land, forced, method, directed)   # :15-16
    fn = CGE.wGCL_directed if directed else CGE.wGCL                                 # :18-24
    results = fn(edges, weights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges, init_eweights,
                 init_embed, split, seed, samples, verbose)
    print(julia_vector(results))                                                     # :25
    return 0


if __name__ == "__main__":
    sys.exit(main())
