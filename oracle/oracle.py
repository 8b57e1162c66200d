"""ctypes front-end of the CPU oracle (oracle/cge_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (cge.jl_amd/) must never import this module.

Signatures mirror the reference's (src/landmarks.jl:365-367, src/divergence.jl:27-31,282-286);
the only additions are the explicit sample arrays (the Julia RNG stream cannot be reproduced,
see the header of cge_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcge_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cge_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class _Samples(C.Structure):
    _fields_ = [("S", C.c_int64), ("n_sets", C.c_int64), ("pos_idx", C.c_void_p), ("neg_i", C.c_void_p),
                ("neg_j", C.c_void_p), ("pos_idx2", C.c_void_p)]


class Trace(C.Structure):
    _fields_ = [("n_alpha", C.c_int64), ("iters", C.c_int64 * 64), ("div", C.c_double * 64),
                ("auc", C.c_double * 64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_idx.restype = C.c_int64
        L.orc_idx.argtypes = [C.c_int64] * 3
        L.orc_js.restype = C.c_double
        L.orc_max_pair_dist.restype = C.c_double
        L.orc_total_rss.restype = C.c_double
        L.orc_unique_rows.restype = C.c_int64
        L.orc_lm_N.restype = C.c_int64
        L.orc_lm_N.argtypes = [C.c_void_p]
        L.orc_lm_n_ledges.restype = C.c_int64
        L.orc_lm_n_ledges.argtypes = [C.c_void_p]
        L.orc_lm_truncated.argtypes = [C.c_void_p]
        L.orc_landmarks_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f(a):  # column-major float64 buffer (Julia layout); returns flat view keeping memory order
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    return a, a.ravel(order="K")


def _i(a):
    a = np.asfortranarray(np.asarray(a, dtype=np.int64))
    return a, a.ravel(order="K")


def _ptr(flat):
    return flat.ctypes.data_as(C.c_void_p)


class OracleError(Exception):
    CODES = {-1: "AssertionError", -2: "Trying to split homogenous cluster",
             -3: "Unexpected empty cluster generated", -4: "out of memory"}

    def __init__(self, rc):
        super().__init__(self.CODES.get(rc, f"error {rc}"))
        self.rc = rc


def idx(n, i, j):
    return lib().orc_idx(n, i, j)


def JS(vC, vB, vI=None, internal=True):
    vC = np.ascontiguousarray(vC, dtype=np.float64)
    vB = np.ascontiguousarray(vB, dtype=np.float64)
    if vI is None or len(vI) == 0:
        vi_p = None
    else:
        vI = np.ascontiguousarray(vI, dtype=np.uint8)
        vi_p = _ptr(vI)
    return lib().orc_js(_ptr(vC), _ptr(vB), C.c_int64(len(vC)), vi_p, C.c_int(1 if internal else 0))


def max_pair_dist(embedding):
    e, ef = _f(embedding)
    return lib().orc_max_pair_dist(_ptr(ef), C.c_int64(e.shape[0]), C.c_int64(e.shape[1]))


def set_known_diameter(hi):
    """Full-size fixtures only: hand the point-set diameter to wGCL / wGCL_directed in landmark mode instead of running
    the O(n^2 d) loop of src/divergence.jl:104-113 (0 = run the loop).  See the comment in cge_oracle.c."""
    lib().orc_set_known_diameter(C.c_double(float(hi)))


def unique_rows(embedding):
    e, ef = _f(embedding)
    return lib().orc_unique_rows(_ptr(ef), C.c_int64(e.shape[0]), C.c_int64(e.shape[1]))


def eig_top(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    v = np.zeros(A.shape[0])
    lib().orc_eig_top(_ptr(A), C.c_int64(A.shape[0]), _ptr(v))
    return v


_EIG_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double))
_eig_cb = None  # keeps the callback object alive while the C side holds its address


def use_lapack_eig(on: bool = True):
    """Fixture generation: take `eigvecs(A)[:, end]` (src/landmarks.jl:99,162,225,254) from LAPACK's syevr -- the routine
    Julia's `eigvecs` calls on a symmetric matrix -- through scipy.linalg.eigh(driver="evr", all eigenpairs as `eigvecs`
    computes them) instead of the oracle's cyclic Jacobi, which needs 26-136 s per 512-wide matrix.  The oracle's sign rule
    (largest-|component| positive) is applied on the C side either way.  tests/test_oracle_golden.py checks that both
    routines give the same partitions on the committed fixtures."""
    global _eig_cb
    if not on:
        lib().orc_set_eig_hook(None)
        _eig_cb = None
        return
    from scipy.linalg import eigh

    def cb(Ap, d, vp):
        try:
            A = np.ctypeslib.as_array(Ap, shape=(d, d))
            w, V = eigh(A, driver="evr", check_finite=False)
            np.ctypeslib.as_array(vp, shape=(d,))[:] = V[:, -1]
            return 0
        except Exception:  # noqa: BLE001 -- nothing may propagate through the C frame
            return 1

    _eig_cb = _EIG_FN(cb)
    lib().orc_set_eig_hook(_eig_cb)


def total_rss(embedding, w, idxs):
    e, ef = _f(embedding)
    w = np.ascontiguousarray(w, dtype=np.float64)
    idxs = np.ascontiguousarray(idxs, dtype=np.int64)
    return lib().orc_total_rss(_ptr(ef), _ptr(w), C.c_int64(e.shape[0]), C.c_int64(e.shape[1]), _ptr(idxs),
                               C.c_int64(len(idxs)))


def split(embedding, w, idxs, method):
    """One application of a split rule to view(embedding, idxs, :); returns 1-based local positions."""
    e, ef = _f(embedding)
    w = np.ascontiguousarray(w, dtype=np.float64)
    idxs = np.ascontiguousarray(idxs, dtype=np.int64)
    k = len(idxs)
    low = np.zeros(max(k, 2), dtype=np.int64)
    high = np.zeros(max(k, 2), dtype=np.int64)
    nl, nh = C.c_int64(0), C.c_int64(0)
    rc = lib().orc_split(_ptr(ef), _ptr(w), C.c_int64(e.shape[0]), C.c_int64(e.shape[1]), _ptr(idxs), C.c_int64(k),
                         C.c_int(_method_code(method)), _ptr(low), C.byref(nl), _ptr(high), C.byref(nh))
    if rc:
        raise OracleError(rc)
    return low[: nl.value] + 1, high[: nh.value] + 1


def _method_code(method):
    if isinstance(method, int):
        return method
    if isinstance(method, str):
        return {"rss": 0, "rss2": 1, "size": 2, "diameter": 3}[method]
    return method.code


def _flatten_clusters(clusters):
    off = np.zeros(len(clusters) + 1, dtype=np.int64)
    for k, c in enumerate(clusters):
        off[k + 1] = off[k] + len(c)
    flat = np.concatenate([np.asarray(c, dtype=np.int64) for c in clusters]) if clusters else np.zeros(0, np.int64)
    return np.ascontiguousarray(flat), off


def runsplit(embedding, w, clusters, n, s, method):
    e, ef = _f(embedding)
    w = np.ascontiguousarray(w, dtype=np.float64)
    flat, off = _flatten_clusters(clusters)
    out = np.zeros(e.shape[0], dtype=np.int64)
    rc = lib().orc_runsplit(_ptr(ef), _ptr(w), C.c_int64(e.shape[0]), C.c_int64(e.shape[1]), _ptr(flat), _ptr(off),
                            C.c_int64(len(clusters)), C.c_int64(n), C.c_int64(s), C.c_int(_method_code(method)),
                            _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def landmarks(edges, weights, vweights, clusters, comm, embedding, verbose, land, forced, method, directed):
    """src/landmarks.jl:365-466 -> (dii, embed, cluster, landmark_edges, weights, lweight, v_to_l)."""
    ed, edf = _i(edges)
    em, emf = _f(embedding)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    vweights = np.ascontiguousarray(vweights, dtype=np.float64)
    cm = np.ascontiguousarray(np.asarray(comm, dtype=np.int64).ravel())
    flat, off = _flatten_clusters(clusters)
    n, d = em.shape
    res = C.c_void_p()
    L = lib()
    rc = L.orc_landmarks(_ptr(edf), C.c_int64(ed.shape[0]), _ptr(weights), _ptr(vweights), C.c_int64(n), _ptr(flat),
                         _ptr(off), C.c_int64(len(clusters)), _ptr(cm), _ptr(emf), C.c_int64(d), C.c_int64(land),
                         C.c_int64(forced), C.c_int(_method_code(method)), C.c_int(1 if directed else 0),
                         C.byref(res))
    if rc:
        raise OracleError(rc)
    try:
        N, ne = L.orc_lm_N(res), L.orc_lm_n_ledges(res)
        dii = np.zeros(N)
        embed = np.zeros((N, d), order="F")
        cluster = np.zeros(N, dtype=np.int64)
        ledges = np.zeros((ne, 2), dtype=np.int64, order="F")
        lw = np.zeros(ne)
        lweight = np.zeros(N)
        v_to_l = np.zeros(n, dtype=np.int64)
        L.orc_lm_get(res, _ptr(dii), _ptr(embed.ravel(order="K")), _ptr(cluster), _ptr(ledges.ravel(order="K")),
                     _ptr(lw), _ptr(lweight), _ptr(v_to_l))
    finally:
        L.orc_landmarks_free(res)
    return dii, embed, cluster.reshape(-1, 1), ledges, lw, lweight, v_to_l


def _samples_struct(samples, keep):
    pos_idx, neg_i, neg_j = [np.ascontiguousarray(np.atleast_2d(a), dtype=np.int64) for a in samples[:3]]
    keep += [pos_idx, neg_i, neg_j]
    s = _Samples()
    s.n_sets, s.S = pos_idx.shape
    s.pos_idx, s.neg_i, s.neg_j = pos_idx.ctypes.data, neg_i.ctypes.data, neg_j.ctypes.data
    s.pos_idx2 = None
    if len(samples) > 3 and samples[3] is not None:
        p2 = np.ascontiguousarray(np.atleast_2d(samples[3]), dtype=np.int64)
        keep.append(p2)
        s.pos_idx2 = p2.ctypes.data
    return s


def _wgcl_common(fn, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                 init_eweights, init_embed, split, samples, want_trace, directed):
    keep = []
    ed, edf = _i(edges)
    em, emf = _f(embed)
    eweights = np.ascontiguousarray(eweights, dtype=np.float64)
    cm = np.ascontiguousarray(np.asarray(comm, dtype=np.int64).ravel())
    distances = np.ascontiguousarray(distances, dtype=np.float64)
    vweights = np.ascontiguousarray(vweights, dtype=np.float64)
    init_vweights = np.ascontiguousarray(init_vweights, dtype=np.float64)
    v_to_l = np.ascontiguousarray(v_to_l, dtype=np.int64)
    ie, ief = _i(init_edges if np.size(init_edges) else np.zeros((0, 2), np.int64))
    init_eweights = np.ascontiguousarray(init_eweights, dtype=np.float64)
    iem, iemf = _f(init_embed if np.size(init_embed) else np.zeros((0, em.shape[1])))
    s = _samples_struct(samples, keep)
    out = np.zeros(7)
    tr = Trace()
    args = [_ptr(edf), C.c_int64(ed.shape[0]), _ptr(eweights), _ptr(cm), C.c_int64(len(cm)), _ptr(emf),
            C.c_int64(em.shape[0]), C.c_int64(em.shape[1]), _ptr(distances), C.c_int64(len(distances)),
            _ptr(vweights), _ptr(init_vweights), C.c_int64(len(init_vweights)), _ptr(v_to_l),
            C.c_int64(len(v_to_l)), _ptr(ief), C.c_int64(ie.shape[0]), _ptr(init_eweights), _ptr(iemf),
            C.c_int(1 if split else 0), C.byref(s), _ptr(out)]
    if directed:
        olen = C.c_int(7)
        rc = fn(*args, C.byref(olen), C.byref(tr))
        n_out = olen.value
    else:
        rc = fn(*args, C.byref(tr))
        n_out = 7
    if rc:
        raise OracleError(rc)
    res = out[:n_out].copy()
    if want_trace:
        k = tr.n_alpha
        return res, {"iters": list(tr.iters[:k]), "div": list(tr.div[:k]), "auc": list(tr.auc[:k])}
    return res


def wGCL(edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges, init_eweights,
         init_embed, split, samples, trace=False):
    """src/divergence.jl:27-257.  `samples` = (pos_idx, neg_i, neg_j), each (n_sets, S), 1-based."""
    return _wgcl_common(lib().orc_wgcl, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l,
                        init_edges, init_eweights, init_embed, split, samples, trace, False)


def wGCL_directed(edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
                  init_eweights, init_embed, split, samples, trace=False):
    """src/divergence.jl:282-561.  `samples` = (pos_idx, neg_i, neg_j[, pos_idx2])."""
    return _wgcl_common(lib().orc_wgcl_directed, edges, eweights, comm, embed, distances, vweights, init_vweights,
                        v_to_l, init_edges, init_eweights, init_embed, split, samples, trace, True)


def louvain_level1(edges, weights, n):
    """Level 1 of Louvain (what src/clustering.jl:14-68 writes to <file>.ecg), nodes visited in the order 0..n-1 (the
    reference's executable shuffles them with an unseeded rand()).  Returns (comm 0-based (n,), n_comm, modularity)."""
    ed, edf = _i(edges)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros(n, dtype=np.int64)
    nc, q = C.c_int64(), C.c_double()
    rc = lib().orc_louvain_level1(_ptr(edf), None if w is None else _ptr(w), C.c_int64(ed.shape[0]), C.c_int64(n), _ptr(out),
                                  C.byref(nc), C.byref(q))
    if rc:
        raise OracleError(rc)
    return out, nc.value, q.value
