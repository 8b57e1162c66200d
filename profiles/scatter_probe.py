"""Times the per-edge cluster-pair scatter alone on the headline graph (for rocprofv3 --kernel-trace --stats):
`python3 profiles/scatter_probe.py [reps] [workload]`; CGE_SCATTER_GATHER=1 selects the gather + atomics kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from cge.jl_amd import api, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "headline"]
directed = bool(wl.get("directed", False))
g = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"] if len(sys.argv) > 3 else 4, seed=42, directed=directed)
ctx = api.Context(0)
ctx.set_graph(g["edges"], g["eweights"], g["n"])
ctx.set_vertex_data(g["comm"], g["vweights"])
C = g["C"]
_, vc = ctx.edge_scatter(None, 1, C, directed, want_wedges=False)
comm = g["comm"][:, 0] - 1
ca, cb = comm[g["edges"][:, 0] - 1], comm[g["edges"][:, 1] - 1]
if directed:
    exp = np.bincount(ca * C + cb, minlength=C * C).astype(float)
else:
    lo, hi = np.minimum(ca, cb), np.maximum(ca, cb)
    exp = np.bincount(C * lo - lo * (lo - 1) // 2 + (hi - lo), minlength=C * (C + 1) // 2).astype(float)
assert os.environ.get("CGE_EB_STOP") or np.array_equal(vc, exp), "scatter differs from numpy"
ctx.profile_select(("edge_scatter",))
ctx.profile_enable(True)
ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.edge_scatter(None, 1, C, directed, want_wedges=False)
t = time.perf_counter() - t0
p = ctx.profile()["edge_scatter"]
ms = p["total_ms"] / p["launches"]
print(f"edge_scatter: {ms * 1e3:.1f} us per pass (events), 24 B x {g['m']} edges -> {24 * g['m'] / ms / 1e6:.0f} GB/s "
      f"= {24 * g['m'] / ms / 1e6 / 8000:.3f} of 8 TB/s; host loop {t / reps * 1e3:.2f} ms per call incl. the D2H of vect_C")
ctx.close()
