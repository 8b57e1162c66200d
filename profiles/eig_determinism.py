import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from cge.jl_amd import api
ctx = api.Context()
rng = np.random.default_rng(1)
for d in (16, 33, 64, 100, 128):
    T = 300
    A = np.empty((T, d, d))
    for t in range(T):
        Y = rng.normal(size=(2 * d, d)) * rng.uniform(0.5, 2.0, size=d)
        A[t] = Y.T @ Y
    v0 = ctx.group_eig(A).copy()
    bad = 0
    for rep in range(5):
        v = ctx.group_eig(A)
        bad += int((v != v0).any(axis=1).sum())
    # rank-deficient ones as well: clusters of 2 .. d/2 points
    B = np.empty((T, d, d))
    for t in range(T):
        m = int(rng.integers(2, max(3, d // 2)))
        Y = rng.normal(size=(m, d))
        Y -= Y.mean(axis=0)
        B[t] = Y.T @ Y
    b0 = ctx.group_eig(B).copy()
    badb = 0
    for rep in range(5):
        badb += int((ctx.group_eig(B) != b0).any(axis=1).sum())
    print(d, "rank-deficient: nondeterministic matrices:", badb)
    w, V = np.linalg.eigh(A)
    ref = V[:, :, -1]
    ref *= np.sign(np.take_along_axis(ref, np.abs(ref).argmax(axis=1)[:, None], 1))
    print(d, "nondeterministic matrices:", bad, "max err vs LAPACK:", np.abs(v0 - ref).max())
