"""cge.jl_amd -- MI355X-native (gfx950) divergence-scoring hot path of CGE.jl.

Host-side mirror of the reference's exported interface for this path
(src/CGE.jl:11-21): ``parseargs``, ``landmarks``, ``wGCL``, ``wGCL_directed``, each a thin
ctypes call into the C-ABI library ``csrc/build/libcge_hip.so`` (declared in include/cge_hip.h).
There is NO CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from .args import (METHODS, ParseError, parseargs, split_cluster_diameter, split_cluster_rss,  # noqa: F401
                   split_cluster_rss2, split_cluster_size)


def __getattr__(name):  # lazy: importing the package must not need the GPU library (parseargs is pure host)
    if name in ("landmarks", "wGCL", "wGCL_directed", "score", "Context", "draw_samples", "library_path",
                "load_library", "CGEError"):
        from . import api

        return getattr(api, name)
    if name == "louvain_clust":  # src/CGE.jl:21
        from .clustering import louvain_clust

        return louvain_clust
    raise AttributeError(name)
