"""ctypes binding of libcge_hip.so + the host-side mirror of the reference's interface for the hot
path: ``landmarks`` (src/landmarks.jl:365-367), ``wGCL`` / ``wGCL_directed`` (src/divergence.jl:27-31,
:282-286) with the reference's positional arguments, and ``score`` = example/CGE_CLI.jl:10-24 on
device-resident inputs.

There is no CPU path here: if the library is missing or no GPU is visible every compute entry point
raises ``CGEError``.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

from .args import _SplitRule

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "csrc", "build", "libcge_hip.so")
_lib = None

CGE_OK = 0
_CODES = {-1: "AssertionError", -2: "ErrorException: Trying to split homogenous cluster",
          -3: "ErrorException: Unexpected empty cluster generated", -4: "HIP error / no GPU",
          -5: "collective hook failed", -6: "out of memory", -7: "bad argument"}


class CGEError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"{_CODES.get(code, 'error')} (code {code}){': ' + msg if msg else ''}")
        self.code = code


class AssertionErrorCGE(CGEError, AssertionError):
    """A reference ``@assert`` fired (src/divergence.jl:50,81,303,363; src/landmarks.jl:93,...)."""


class WgclArgs(C.Structure):
    _fields_ = [("edges_src", C.c_void_p), ("edges_dst", C.c_void_p), ("eweights", C.c_void_p), ("m", C.c_int64),
                ("comm", C.c_void_p), ("n_comm", C.c_int64), ("embed", C.c_void_p), ("embed_rows", C.c_int64),
                ("d", C.c_int64), ("distances", C.c_void_p), ("n_distances", C.c_int64), ("vweights", C.c_void_p),
                ("init_vweights", C.c_void_p), ("n_init", C.c_int64), ("v_to_l", C.c_void_p),
                ("n_v_to_l", C.c_int64), ("init_edges_src", C.c_void_p), ("init_edges_dst", C.c_void_p),
                ("m_init", C.c_int64), ("init_eweights", C.c_void_p), ("init_embed", C.c_void_p), ("split", C.c_int),
                ("seed", C.c_int64), ("auc_samples", C.c_int64), ("verbose", C.c_int), ("directed", C.c_int),
                ("pos_idx", C.c_void_p), ("neg_i", C.c_void_p), ("neg_j", C.c_void_p), ("pos_idx2", C.c_void_p),
                ("n_sample_sets", C.c_int64)]


class Trace(C.Structure):
    _fields_ = [("n_alpha", C.c_int64), ("iters", C.c_int64 * 64), ("div", C.c_double * 64),
                ("auc", C.c_double * 64)]

    def as_dict(self):
        k = self.n_alpha
        return {"n_alpha": k, "iters": list(self.iters[:k]), "div": list(self.div[:k]), "auc": list(self.auc[:k])}


class ScoreArgs(C.Structure):
    _fields_ = [("clusters_flat", C.c_void_p), ("clusters_off", C.c_void_p), ("n_clusters", C.c_int64),
                ("land", C.c_int64), ("forced", C.c_int64), ("method", C.c_int), ("directed", C.c_int),
                ("split", C.c_int), ("seed", C.c_int64), ("auc_samples", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)


class Collectives(C.Structure):
    _fields_ = [("allreduce_f64", ALLREDUCE_FN), ("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int)]


GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)  # (user, buf, words per rank): all-gather / reduce-scatter


class CollectivesExt(C.Structure):
    _fields_ = [("allgather", GATHER_FN), ("reduce_scatter_f64", GATHER_FN)]


# Live contexts are closed from an `atexit` hook: it runs inside Py_Finalize, i.e. BEFORE the C-level exit handlers
# of the HIP runtime and of a profiler's tool library, so no stream or event of this library is left for the runtime to
# destroy after a profiler has finalised its HSA hooks.  DEFENCE IN DEPTH ONLY: it is NOT an established fix for the
# exit-time SIGSEGV under rocprofv3 recorded in round 1 -- a probe that left a context alive with this hook disabled exited
# cleanly, so the cause of that fault is still unproven (the cooperative launches that build used
# are gone; the fault has not reappeared in any record since).
_live = weakref.WeakSet()
_atexit_registered = False


def _close_live_contexts():
    for ctx in list(_live):
        try:
            ctx.close()
        except Exception:
            pass


def library_path():
    return _LIB_PATH


def load_library():
    """Load libcge_hip.so; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise CGEError(-4, f"{_LIB_PATH} not found: build it with `make -C cge.jl_amd/csrc` "
                               f"(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(_LIB_PATH)
        L.cge_last_error.restype = C.c_char_p
        L.cge_last_error.argtypes = [C.c_void_p]
        L.cge_idx.restype = C.c_int64
        L.cge_idx.argtypes = [C.c_int64] * 3
        L.cge_destroy.argtypes = [C.c_void_p]
        L.cge_destroy.restype = None
        _lib = L
    return _lib


def read_table(path, column_major=True, n_threads=0):
    """Numeric text table -> float64 array (rows, cols) through the library's parallel reader
    (include/cge_hip.h: cge_text_table_*; replaces `readdlm`, src/auxilary.jl:80-168).  Needs no GPU.
    Returns (array, header_skipped)."""
    L = load_library()
    L.cge_text_table_close.argtypes = [C.c_void_p]
    L.cge_text_table_close.restype = None
    rows, cols, hdr, h = C.c_int64(), C.c_int64(), C.c_int(), C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.cge_text_table_open(os.fsencode(path), C.c_int(n_threads), C.byref(rows), C.byref(cols), C.byref(hdr),
                               C.byref(h), err, C.c_int64(512))
    if rc != 0:
        raise CGEError(rc, err.value.decode() or f"cannot read {path}")
    try:
        out = np.empty((rows.value, cols.value), dtype=np.float64, order="F" if column_major else "C")
        rc = L.cge_text_table_parse(h, out.ctypes.data_as(C.c_void_p), C.c_int(1 if column_major else 0), err,
                                    C.c_int64(512))
        if rc != 0:
            raise CGEError(rc, err.value.decode() or f"cannot parse {path}")
    finally:
        L.cge_text_table_close(h)
    return out, bool(hdr.value)


def _method_code(method):
    if isinstance(method, _SplitRule):
        return method.code
    if isinstance(method, str):
        return {"rss": 0, "rss2": 1, "size": 2, "diameter": 3}[method]
    return int(method)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _colmajor(a):  # (rows, cols) array -> flat buffer in Julia (column-major) order
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    return a, a.ravel(order="K")


def _edge_cols(edges):
    e = np.asarray(edges, dtype=np.int64)
    if e.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])


class FlatClusters:
    """`clusters::Vector{Vector{Int}}` (src/auxilary.jl:199-208) in the form the C-ABI takes: member ids back to back + offsets.
    Build it once with `flatten_clusters` and hand it to `Context.score` / `landmarks_run` instead of the list of lists when
    the same clusters are scored repeatedly (flattening a million ids is ~1 ms of numpy per call)."""

    def __init__(self, flat, off):
        self.flat, self.off = flat, off

    def __len__(self):
        return len(self.off) - 1


def flatten_clusters(clusters):
    return FlatClusters(*_flatten_clusters_raw(clusters))


_flat_cache = {}  # id(list) -> (fingerprint, FlatClusters): the last few lists of clusters seen, by identity


def _flatten_clusters(clusters):
    """Flat form of a list of clusters.  A list object that was flattened before is recognised by identity (the list and every
    member array: ids and lengths) and its flat form reused -- callers that score the same clusters again and again (bench.py,
    a sweep over embeddings) then pay the ~1 ms of concatenation once.  Contract: do not change a cluster's members IN PLACE
    between calls; build a new list (or a FlatClusters) instead."""
    if isinstance(clusters, FlatClusters):
        return clusters.flat, clusters.off
    if isinstance(clusters, list) and len(clusters) > 16:
        fp = (len(clusters), tuple((id(c), len(c)) for c in clusters))
        hit = _flat_cache.get(id(clusters))
        if hit is not None and hit[0] == fp:
            return hit[1].flat, hit[1].off
        fc = flatten_clusters(clusters)
        if len(_flat_cache) >= 4:
            _flat_cache.pop(next(iter(_flat_cache)))
        _flat_cache[id(clusters)] = (fp, fc, clusters)  # (the list is kept alive: its id cannot be reused while the entry lives)
        return fc.flat, fc.off
    return _flatten_clusters_raw(clusters)


def _flatten_clusters_raw(clusters):
    off = np.zeros(len(clusters) + 1, dtype=np.int64)
    for k, c in enumerate(clusters):
        off[k + 1] = off[k] + len(c)
    flat = np.concatenate([np.asarray(c, dtype=np.int64) for c in clusters]) if len(clusters) else np.zeros(0, np.int64)
    return np.ascontiguousarray(flat), off


class Context:
    """One GPU, one stream.  Not re-entrant (one host thread drives it)."""

    def __init__(self, device: int = 0, stream=None):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.cge_create(C.byref(h), C.c_int(device), C.c_void_p(stream) if stream else None)
        if rc:
            raise CGEError(rc, "cge_create failed (no MI355X visible?)")
        self.h = h
        self.device = device
        self._keep = []  # objects the C side holds pointers to (collective hook)
        self.n = self.m = self.d = 0
        global _atexit_registered
        _live.add(self)
        if not _atexit_registered and not os.environ.get("CGE_NO_ATEXIT_CLOSE"):
            atexit.register(_close_live_contexts)
            _atexit_registered = True

    def close(self):
        if getattr(self, "h", None):
            self.L.cge_destroy(self.h)
            self.h = None
        _live.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            msg = self.L.cge_last_error(self.h).decode(errors="replace")
            raise (AssertionErrorCGE if rc == -1 else CGEError)(rc, msg)

    # ---- resident inputs ------------------------------------------------------------------------
    def set_host_threads(self, n):
        self._check(self.L.cge_set_host_threads(self.h, C.c_int(n)))

    def set_graph(self, edges, eweights, n):
        s, t = _edge_cols(edges)
        w = _f64(eweights)
        self._check(self.L.cge_set_graph(self.h, _p(s), _p(t), _p(w), C.c_int64(len(s)), C.c_int64(n)))
        self.m, self.n = len(s), n

    def set_embedding(self, embedding):
        e, ef = _colmajor(embedding)
        self._check(self.L.cge_set_embedding(self.h, _p(ef), C.c_int64(e.shape[0]), C.c_int64(e.shape[1])))
        self.d = e.shape[1]

    def set_embedding_device(self, dev_ptr: int, n: int, d: int, row_major: bool = True):
        """An (n, d) float64 embedding that already lives in this GPU's memory (e.g. `tensor.data_ptr()`)."""
        self._check(self.L.cge_set_embedding_device(self.h, C.c_void_p(dev_ptr), C.c_int64(n), C.c_int64(d),
                                                    C.c_int(1 if row_major else 0)))
        self.d = d

    def set_vertex_data(self, comm, vweights):
        cm = None if comm is None else _i64(np.asarray(comm).ravel())
        vw = None if vweights is None else _f64(vweights)
        n = len(cm) if cm is not None else len(vw)
        self._check(self.L.cge_set_vertex_data(self.h, _p(cm), _p(vw), C.c_int64(n)))

    def set_inputs(self, edges, eweights, vweights, comm, embedding):
        n = int(np.asarray(embedding).shape[0])
        self.set_graph(edges, eweights, n)
        self.set_vertex_data(comm, vweights)  # before the embedding: option shard_rows shards the rows BY COMMUNITY
        self.set_embedding(embedding)

    # ---- landmarks ----------------------------------------------------------------------------------
    def landmarks_run(self, clusters, land, forced, method, directed=False):
        flat, off = _flatten_clusters(clusters)
        N, ne, tr = C.c_int64(), C.c_int64(), C.c_int()
        self._check(self.L.cge_landmarks_run(self.h, _p(flat), _p(off), C.c_int64(len(clusters)), C.c_int64(land),
                                             C.c_int64(forced), C.c_int(_method_code(method)),
                                             C.c_int(1 if directed else 0), C.byref(N), C.byref(ne), C.byref(tr)))
        self.N, self.n_ledges, self.truncated = N.value, ne.value, bool(tr.value)
        return self.N, self.n_ledges, self.truncated

    def landmarks_info(self):
        N, ne, tr = C.c_int64(), C.c_int64(), C.c_int()
        self._check(self.L.cge_landmarks_info(self.h, C.byref(N), C.byref(ne), C.byref(tr)))
        self.N, self.n_ledges, self.truncated = N.value, ne.value, bool(tr.value)
        return self.N, self.n_ledges, self.truncated

    def landmarks_fetch(self):
        self.landmarks_info()  # sizes come from the library, never from Python-side state
        N, ne, n, d = self.N, self.n_ledges, self.n, self.d
        dii = np.zeros(N)
        embed = np.zeros((N, d), order="F")
        cluster = np.zeros(N, dtype=np.int64)
        ledges = np.zeros((ne, 2), dtype=np.int64, order="F")
        lw = np.zeros(ne)
        lweight = np.zeros(N)
        v_to_l = np.zeros(n, dtype=np.int64)
        self._check(self.L.cge_landmarks_fetch(self.h, _p(dii), _p(embed.ravel(order="K")), _p(cluster),
                                               _p(ledges.ravel(order="K")), _p(lw), _p(lweight), _p(v_to_l)))
        return dii, embed, cluster.reshape(-1, 1), ledges, lw, lweight, v_to_l

    def runsplit(self, clusters, nland, forced, method):
        flat, off = _flatten_clusters(clusters)
        out = np.zeros(self.n, dtype=np.int64)
        self._check(self.L.cge_runsplit(self.h, _p(flat), _p(off), C.c_int64(len(clusters)), C.c_int64(nland),
                                        C.c_int64(forced), C.c_int(_method_code(method)), _p(out)))
        return out

    # ---- samples ---------------------------------------------------------------------------------------
    def draw_samples(self, seed, S, directed=False, stream_id=0):
        pos = np.zeros(S, dtype=np.int64)
        ni = np.zeros(S, dtype=np.int64)
        nj = np.zeros(S, dtype=np.int64)
        self._check(self.L.cge_draw_samples(self.h, C.c_int64(seed), C.c_int64(stream_id), C.c_int64(S),
                                            C.c_int(1 if directed else 0), _p(pos), _p(ni), _p(nj)))
        return pos, ni, nj

    # ---- wGCL ------------------------------------------------------------------------------------------
    def wgcl(self, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
             init_eweights, init_embed, split, seed=-1, auc_samples=10000, verbose=False, directed=False,
             samples=None, use_resident_original=False):
        keep = []
        s, t = _edge_cols(edges)
        ew = _f64(eweights)
        cm = _i64(np.asarray(comm).ravel())
        em, emf = _colmajor(embed)
        dist = _f64(distances)
        vw = _f64(vweights)
        ivw = _f64(init_vweights)
        v2l = _i64(v_to_l)
        a = WgclArgs()
        a.edges_src, a.edges_dst, a.eweights, a.m = _p(s).value, _p(t).value, _p(ew).value, len(s)
        a.comm, a.n_comm = _p(cm).value, len(cm)
        a.embed, a.embed_rows, a.d = _p(emf).value, em.shape[0], em.shape[1]
        a.distances, a.n_distances = _p(dist).value, len(dist)
        a.vweights = _p(vw).value
        a.init_vweights, a.n_init = (_p(ivw).value if len(ivw) else None), len(ivw)
        a.v_to_l, a.n_v_to_l = (_p(v2l).value if len(v2l) else None), len(v2l)
        if len(v2l) and not use_resident_original:
            is_, it_ = _edge_cols(init_edges)
            iew = _f64(init_eweights)
            iem, iemf = _colmajor(init_embed)
            keep += [is_, it_, iew, iem, iemf]
            a.init_edges_src, a.init_edges_dst, a.m_init = _p(is_).value, _p(it_).value, len(is_)
            a.init_eweights, a.init_embed = _p(iew).value, _p(iemf).value
        a.split, a.seed, a.auc_samples = int(bool(split)), int(seed), int(auc_samples)
        a.verbose, a.directed = int(bool(verbose)), int(bool(directed))
        if samples is not None:
            arrs = [np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64) for x in samples[:3]]
            keep += arrs
            a.pos_idx, a.neg_i, a.neg_j = (_p(x).value for x in arrs)
            a.n_sample_sets, a.auc_samples = arrs[0].shape
            if len(samples) > 3 and samples[3] is not None:
                p2 = np.ascontiguousarray(np.atleast_2d(samples[3]), dtype=np.int64)
                keep.append(p2)
                a.pos_idx2 = _p(p2).value
        out = np.zeros(7)
        olen = C.c_int(7)
        tr = Trace()
        self._check(self.L.cge_wgcl(self.h, C.byref(a), _p(out), C.byref(olen), C.byref(tr)))
        self.last_trace = tr.as_dict()
        del keep
        return out[: olen.value].copy()

    def score(self, clusters, land, forced=4, method="rss", directed=False, split=False, seed=-1, auc_samples=10000):
        """example/CGE_CLI.jl:10-24 on the resident inputs; land = -1 => exact mode."""
        flat, off = _flatten_clusters(clusters if clusters else [])
        a = ScoreArgs()
        a.clusters_flat, a.clusters_off, a.n_clusters = _p(flat).value, _p(off).value, len(off) - 1
        a.land, a.forced, a.method = int(land), int(forced), _method_code(method)
        a.directed, a.split, a.seed, a.auc_samples = int(bool(directed)), int(bool(split)), int(seed), int(auc_samples)
        out = np.zeros(7)
        olen = C.c_int(7)
        tr = Trace()
        self._check(self.L.cge_score(self.h, C.byref(a), _p(out), C.byref(olen), C.byref(tr)))
        self.last_trace = tr.as_dict()
        return out[: olen.value].copy()

    # ---- kernel-level ----------------------------------------------------------------------------------
    def edge_scatter(self, v_to_l, N, Cn, directed=False, e0=0, e1=None, want_wedges=True, want_vect_c=True):
        e1 = self.m if e1 is None else e1
        v2l = None if v_to_l is None else _i64(v_to_l)
        wed = np.zeros((N, N)) if want_wedges else None
        vlen = Cn * Cn if directed else Cn * (Cn + 1) // 2
        vc = np.zeros(vlen) if want_vect_c else None
        self._check(self.L.cge_edge_scatter(self.h, _p(v2l), C.c_int64(N), C.c_int64(Cn), C.c_int(int(directed)),
                                            C.c_int64(e0), C.c_int64(e1), _p(wed), _p(vc)))
        return wed, vc

    def max_pair_dist(self, part=0, nparts=1):
        hi, ai, aj = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self.L.cge_max_pair_dist(self.h, C.c_int(part), C.c_int(nparts), C.byref(hi), C.byref(ai),
                                             C.byref(aj)))
        return hi.value, ai.value, aj.value

    def group_eig(self, A):
        """Testing hook (include/cge_hip_testing.h): principal eigenvectors of a (T, d, d) stack of symmetric
        matrices by the batched device solver of the landmark phase (replaces `eigvecs(A)[:, end]`)."""
        A = _f64(A)
        T, d = A.shape[0], A.shape[1]
        v = np.empty((T, d), dtype=np.float64)
        self._check(self.L.cge_group_eig(self.h, _p(A), C.c_int64(T), C.c_int64(d), _p(v)))
        return v

    def pow_test(self, x, alpha, method):
        """Testing hook: (1 - x)^alpha on the device; method 0 = library pow, 1 = the sweep's exp2(alpha * log2(1 - x))."""
        x = _f64(x)
        out = np.empty_like(x)
        self._check(self.L.cge_pow_test(self.h, _p(x), C.c_int64(x.size), C.c_double(alpha), C.c_int(method), _p(out)))
        return out

    def segment_sort_test(self, z, offsets):
        """Testing hook: the per-group stable sort of runsplit's projections; returns (sorted z, local permutation)."""
        z = _f64(z)
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        zs, perm = np.empty_like(z), np.empty(z.size, dtype=np.int32)
        self._check(self.L.cge_segment_sort_test(self.h, _p(z), _p(off), C.c_int64(off.size - 1), _p(zs), _p(perm)))
        return zs, perm

    def wave_tree_test(self, x):
        """Testing hook: per row of 64 doubles the shuffle-tree sum and the lane-swap sum of the projection kernel."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 64)
        a, b = np.empty(x.shape[0]), np.empty(x.shape[0])
        self._check(self.L.cge_wave_tree_test(self.h, _p(x), C.c_int64(x.shape[0]), _p(a), _p(b)))
        return a, b

    def js(self, vC, vB, vI=None, internal=True):
        vC, vB = _f64(vC), _f64(vB)
        vi = None if vI is None or len(vI) == 0 else np.ascontiguousarray(vI, dtype=np.uint8)
        out = C.c_double()
        self._check(self.L.cge_js(self.h, _p(vC), _p(vB), C.c_int64(len(vC)), _p(vi), C.c_int(int(bool(internal))),
                                  C.byref(out)))
        return out.value

    # ---- collectives / profiling -----------------------------------------------------------------------
    def exchange_buffer(self, min_doubles):
        ptr, cap = C.c_void_p(), C.c_int64()
        self._check(self.L.cge_exchange_buffer(self.h, C.c_int64(min_doubles), C.byref(ptr), C.byref(cap)))
        return ptr.value, cap.value

    def set_collectives(self, fn, rank, world):
        cb = ALLREDUCE_FN(fn)
        coll = Collectives(cb, None, rank, world)
        self._keep = [cb, coll]
        self._check(self.L.cge_set_collectives(self.h, C.byref(coll)))

    def set_collectives_ext(self, allgather=None, reduce_scatter=None):
        """Optional further ops of the hook (include/cge_hip.h: cge_collectives_ext); call after set_collectives."""
        ag = GATHER_FN(allgather) if allgather else GATHER_FN()
        rs = GATHER_FN(reduce_scatter) if reduce_scatter else GATHER_FN()
        ext = CollectivesExt(ag, rs)
        self._keep = list(getattr(self, "_keep", [])) + [ag, rs, ext]
        self._check(self.L.cge_set_collectives_ext(self.h, C.byref(ext)))

    def clear_collectives(self):
        """Back to a single-rank context: the in-library communicator is released and the hook removed."""
        self._check(self.L.cge_comm_finalize(self.h))
        self._check(self.L.cge_set_collectives(self.h, None))
        self._keep = []

    def init_rccl(self, unique_id: bytes, rank: int, world: int):
        """In-library collectives (include/cge_hip.h: cge_comm_init_rccl): every rank calls this with rank 0's id."""
        assert len(unique_id) == 128
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.L.cge_comm_init_rccl(self.h, buf, C.c_int(rank), C.c_int(world)))

    def finalize_rccl(self):
        """Release the in-library communicator (cge_comm_finalize); collectives go back to the hook, if one is set."""
        self._check(self.L.cge_comm_finalize(self.h))

    def rccl_selftest(self, arr, op=0):
        """Testing hook: all-reduce `arr` (float64, or int64 for op 2) through the context's communicator."""
        a = np.ascontiguousarray(arr).copy()
        assert a.dtype.itemsize == 8
        self._check(self.L.cge_rccl_selftest(self.h, _p(a), C.c_int64(a.size), C.c_int(op)))
        return a

    def louvain(self):
        """Level-1 Louvain communities of the resident graph: (comm 0-based (n,), n_comm, modularity, rounds)."""
        out = np.zeros(self.n, dtype=np.int64)
        nc, q, rounds = C.c_int64(), C.c_double(), C.c_int64()
        self._check(self.L.cge_louvain(self.h, _p(out), C.byref(nc), C.byref(q), C.byref(rounds)))
        return out, nc.value, q.value, rounds.value

    _TEST_OPTIONS = ("fit_persistent_test_delay", "fit_persistent_test_timeout", "test_bvec_plain")

    def set_option(self, key, value):
        if key in self._TEST_OPTIONS:  # the testing knobs are not part of the boundary (include/cge_hip_testing.h)
            self._check(self.L.cge_set_test_option(self.h, key.encode(), C.c_int64(int(value))))
            return
        self._check(self.L.cge_set_option(self.h, key.encode(), C.c_int64(int(value))))

    def get_stat(self, key):
        v = C.c_int64()
        self._check(self.L.cge_get_stat(self.h, key.encode(), C.byref(v)))
        return v.value

    def last_diameter(self):
        """(hi, path, candidate landmark pairs, candidate tiles) of the last landmark-mode run."""
        import struct

        hi = struct.unpack("d", struct.pack("q", self.get_stat("diameter_bits")))[0]
        return hi, {1: "brute", 2: "pruned"}.get(self.get_stat("diameter_path"), "none"), \
            self.get_stat("diameter_candidate_pairs"), self.get_stat("diameter_candidate_tiles")

    def profile_enable(self, on=True):
        self._check(self.L.cge_profile_enable(self.h, C.c_int(int(on))))

    def profile_select(self, names=()):
        """Time only the named kernels (empty: all) -- every timer is a pair of events on the stream."""
        self._check(self.L.cge_profile_select(self.h, ",".join(names).encode()))

    def profile_reset(self):
        self._check(self.L.cge_profile_reset(self.h))

    def profile(self):
        buf = C.create_string_buffer(4096)
        self._check(self.L.cge_profile_names(self.h, buf, C.c_int64(4096)))
        res = {}
        for name in filter(None, buf.value.decode().split(",")):
            n, ms = C.c_int64(), C.c_double()
            self._check(self.L.cge_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
            res[name] = {"launches": n.value, "total_ms": ms.value}
        return res

    def phase_ms(self):
        res = {}
        buf = C.create_string_buffer(4096)
        self._check(self.L.cge_phase_names(self.h, buf, C.c_int64(4096)))
        for ph in filter(None, buf.value.decode().split(",")):
            ms = C.c_double()
            self._check(self.L.cge_phase_ms(self.h, ph.encode(), C.byref(ms)))
            res[ph] = ms.value
        return res


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def rccl_unique_id() -> bytes:
    """Rank 0: the 128-byte id every rank passes to Context.init_rccl (include/cge_hip.h: cge_rccl_unique_id)."""
    buf = C.create_string_buffer(128)
    rc = load_library().cge_rccl_unique_id(buf)
    if rc:
        raise CGEError(rc, "cge_rccl_unique_id failed (is librccl installed and a GPU visible?)")
    return buf.raw


def idx(n, i, j):
    return load_library().cge_idx(n, i, j)


# ---- the reference's exported functions ------------------------------------------------------------------
def landmarks(edges, weights, vweights, clusters, comm, embedding, verbose, land, forced, method, directed,
              ctx: Context | None = None):
    """landmarks(edges, weights, vweights, clusters, comm, embedding, verbose, land, forced, method, directed)
    -> (dii, embed, cluster, landmark_edges, weights, lweight, v_to_l)        (src/landmarks.jl:365-367, :465)"""
    ctx = ctx or default_context()
    verbose and print("Starts landmark generation")
    ctx.set_inputs(edges, weights, vweights, comm, embedding)
    N, _, truncated = ctx.landmarks_run(clusters, land, forced, method, directed)
    if truncated:
        print("Warning: Requested number of clusters larger than unique no. embeddings. Truncating.", file=sys.stderr)
    verbose and print("Landmarks generated")
    verbose and print(f"Using {N} landmarks")
    return ctx.landmarks_fetch()


def _wgcl(directed, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
          init_eweights, init_embed, split, seed, auc_samples, verbose, samples, trace, ctx):
    ctx = ctx or default_context()
    if verbose:  # the reference's own lines, in its order (src/divergence.jl:43-52,75 / :296-305,354)
        e = np.asarray(edges)
        ie = np.asarray(init_edges) if init_edges is not None else np.zeros((0, 2), np.int64)
        print(f"auc_samples: {auc_samples}")
        print(f"Graph has {int(e.max())} vertices and {e.shape[0]} edges")
        if v_to_l is not None and len(v_to_l) > 0 and ie.size:
            print(f"Original graph has {int(ie.max())} vertices and {ie.shape[0]} edges")
        print(f"Graph has {int(np.asarray(comm).max())} communities")
        print(f"Embedding has {np.asarray(embed).shape[1]} dimensions")
    res = ctx.wgcl(edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                   init_eweights, init_embed, split, seed, auc_samples, verbose, directed, samples)
    sys.stderr.write("." * ctx.last_trace["n_alpha"] + "\n")  # src/divergence.jl:140,255
    return (res, ctx.last_trace) if trace else res


def wGCL(edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges, init_eweights,
         init_embed, split, seed=-1, auc_samples=10000, verbose=False, *, samples=None, trace=False, ctx=None):
    """src/divergence.jl:27-31.  `samples` (optional) = pre-drawn (pos_idx, neg_i, neg_j), each (n_sets, S)."""
    return _wgcl(False, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                 init_eweights, init_embed, split, seed, auc_samples, verbose, samples, trace, ctx)


def wGCL_directed(edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                  init_eweights, init_embed, split, seed=-1, auc_samples=10000, verbose=False, *, samples=None,
                  trace=False, ctx=None):
    """src/divergence.jl:282-286."""
    return _wgcl(True, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                 init_eweights, init_embed, split, seed, auc_samples, verbose, samples, trace, ctx)


def score(edges, eweights, vweights, comm, clusters, embedding, land, forced=4, method="rss", directed=False,
          split=False, seed=-1, auc_samples=10000, ctx=None):
    ctx = ctx or default_context()
    ctx.set_inputs(edges, eweights, vweights, comm, embedding)
    return ctx.score(clusters, land, forced, method, directed, split, seed, auc_samples)


def draw_samples(ctx, seed, S, directed=False, n_sets=1):
    sets = [ctx.draw_samples(seed, S, directed, t) for t in range(n_sets)]
    return tuple(np.stack([s[k] for s in sets]) for k in range(3))
