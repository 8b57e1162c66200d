"""Stage split of the batched eigen-solver (kernels_lm.hip: group_eig_kernel) on T covariance matrices of width d.
Run once per stage: CGE_EIG_DIAG=1 (stop after the tridiagonalisation), 2 (+ multisection), 3 (+ inverse iteration),
0/unset (everything).  Prints the HIP-event time.  (The forms CGE_EIG_FORM selected in rounds 3-4 were removed in round 5.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api

T, d = int(sys.argv[1]) if len(sys.argv) > 1 else 600, int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
A = np.empty((T, d, d))
for t in range(T):
    Y = rng.normal(size=(2 * d, d)) * rng.uniform(0.5, 2.0, size=d)
    A[t] = Y.T @ Y
ctx = api.Context()
ctx.profile_enable(True)
ctx.group_eig(A)
ctx.profile_reset()
for _ in range(5):
    ctx.group_eig(A)
pr = ctx.profile()["group_eig"]
n, ms = pr["launches"], pr["total_ms"]
print(f"CGE_EIG_DIAG={os.environ.get('CGE_EIG_DIAG', '0')} T={T} d={d}: "
      f"{ms / n:.3f} ms per launch")
