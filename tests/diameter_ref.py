"""Exact point-set diameter on the CPU at sizes where the reference's O(n^2 d) loop (`extrema(full_graph_D)`,
src/divergence.jl:104-113; oracle/cge_oracle.c: orc_max_pair_dist) cannot run -- TEST INFRASTRUCTURE, numpy only.

An independent branch and bound (it shares no code with the library's diameter_host.cpp): for any grouping of the
points with group means mu_g and residuals u_i = x_i - mu_a (i in group a), v_j = x_j - mu_b (j in b),

    |x_i - x_j|^2 = |x_i - mu_b|^2 + |x_j - mu_a|^2 - |mu_a - mu_b|^2 - 2 u_i.v_j
                 <= |x_i - mu_b|^2 + Q[b][a] - |mu_a - mu_b|^2 + 2 |u_i| R[b]            (*)

with Q[b][a] = max_{j in b} |x_j - mu_a|^2 and R[b] = max_{j in b} |v_j|.  Pass 1 computes Q, R and, from the extreme
points of every group pair, a lower bound L; pass 2 keeps, for every ordered group pair (a, b), only the points of a
whose bound (*) reaches L; the surviving point sets are compared exhaustively.  The winner is re-evaluated with
dist()'s own arithmetic (src/auxilary.jl:14-20: a sequential sum of squares, then sqrt), together with every pair whose
BLAS-evaluated distance is within 1e-9 of the maximum, so the returned value has the bits of the reference's `hi`.
All floating-point bounds carry a relative safety margin of 1e-9, far above the rounding of the expanded products."""
import numpy as np

MARGIN = 1e-9


def dist_seq(a, b):
    """dist(v1, v2, embed) of src/auxilary.jl:14-20 for rows a, b (vectorised over leading axes, sequential over d)."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    s = np.zeros(a.shape[0])
    for k in range(a.shape[1]):
        t = a[:, k] - b[:, k]
        s = s + t * t
    return np.sqrt(s)


def _sqdist(X, xn, M, mn):
    """|x_i - m_r|^2 by the expanded product (rows of X against rows of M)."""
    return np.maximum(xn[:, None] + mn[None, :] - 2.0 * (X @ M.T), 0.0)


def exact_diameter(X, labels, chunk=65536):
    """(hi, i, j): the largest pairwise Euclidean distance among the rows of X (n, d) and a pair attaining it
    (0-based).  `labels` (n,) is any grouping of the rows (communities); it only affects the pruning."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    _, lab = np.unique(np.asarray(labels), return_inverse=True)
    G = int(lab.max()) + 1
    cnt = np.bincount(lab, minlength=G).astype(np.float64)
    M = np.zeros((G, d))
    np.add.at(M, lab, X)
    M /= cnt[:, None]
    xn = np.einsum("ij,ij->i", X, X)
    mn = np.einsum("ij,ij->i", M, M)
    D2 = np.maximum(mn[:, None] + mn[None, :] - 2.0 * (M @ M.T), 0.0)  # |mu_a - mu_b|^2
    # pass 1: Q[b][a] = max over j in b of |x_j - mu_a|^2, its arg-max, R[b]
    Q = np.full((G, G), -1.0)
    QA = np.zeros((G, G), dtype=np.int64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        P = _sqdist(X[s:e], xn[s:e], M, mn)  # (rows, G)
        lb = lab[s:e]
        order = np.argsort(lb, kind="stable")
        bounds = np.flatnonzero(np.diff(lb[order])) + 1
        for grp in np.split(order, bounds):
            b = lb[grp[0]]
            blk = P[grp]
            am = blk.argmax(0)
            mx = blk[am, np.arange(G)]
            upd = mx > Q[b]
            Q[b][upd] = mx[upd]
            QA[b][upd] = s + grp[am[upd]]
    R = np.sqrt(np.diag(Q))
    # lower bound from the extreme points of every group pair: i* = argmax_{i in a}|x_i - mu_b|, j* likewise
    ia, ib = np.triu_indices(G)
    pi, pj = QA[ia, ib], QA[ib, ia]
    cand = np.maximum(xn[pi] + xn[pj] - 2.0 * np.einsum("ij,ij->i", X[pi], X[pj]), 0.0)
    k = int(cand.argmax())
    best = float(dist_seq(X[pi[k]], X[pj[k]])[0])
    best_pair = (int(pi[k]), int(pj[k]))
    L2 = best * best * (1.0 - MARGIN)
    # group pairs that can still hold a longer pair: Q[a][b] + Q[b][a] - D2 + 2 R[a] R[b] >= L^2
    U = Q + Q.T - D2 + 2.0 * np.outer(R, R)
    alive = U >= L2
    # pass 2: per point i (group a) and group b: |x_i - mu_b|^2 + Q[b][a] - D2[a][b] + 2 |u_i| R[b] >= L^2 ?
    keep_rows, keep_grp = [], []
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        P = _sqdist(X[s:e], xn[s:e], M, mn)
        lb = lab[s:e]
        un = np.sqrt(P[np.arange(e - s), lb])  # |u_i|
        ub = P + Q.T[lb] - D2[lb] + 2.0 * un[:, None] * R[None, :]
        hit = (ub >= L2) & alive[lb]
        r, g = np.nonzero(hit)
        keep_rows.append(r + s)
        keep_grp.append(g)
    rows = np.concatenate(keep_rows)
    grps = np.concatenate(keep_grp)
    # candidates of the ordered group pair (a, b): rows of a that may pair with something in b
    key = lab[rows] * G + grps
    order = np.argsort(key, kind="stable")
    rows, key = rows[order], key[order]
    starts = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    ends = np.concatenate([starts[1:], [len(key)]])
    seg = {int(key[s]): (s, e) for s, e in zip(starts, ends)}
    n_eval = 0
    for kk, (s, e) in seg.items():
        a, b = divmod(kk, G)
        if a > b or (b * G + a) not in seg:
            continue
        s2, e2 = seg[b * G + a]
        A, B = rows[s:e], rows[s2:e2]
        n_eval += len(A) * len(B)
        for t in range(0, len(A), 4096):
            At = A[t:t + 4096]
            d2 = xn[At][:, None] + xn[B][None, :] - 2.0 * (X[At] @ X[B].T)
            m = d2.max()
            if m < L2:
                continue
            ii, jj = np.nonzero(d2 >= m * (1.0 - MARGIN))
            ex = dist_seq(X[At[ii]], X[B[jj]])
            q = int(ex.argmax())
            if ex[q] > best and At[ii[q]] != B[jj[q]]:
                best = float(ex[q])
                best_pair = (int(At[ii[q]]), int(B[jj[q]]))
                L2 = best * best * (1.0 - MARGIN)
    i, j = sorted(best_pair)
    return best, i, j, {"groups": G, "alive_group_pairs": int(np.triu(alive).sum()), "pairs_evaluated": int(n_eval)}
