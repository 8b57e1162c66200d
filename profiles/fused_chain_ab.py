"""A/B aid (round 5): the alpha chain riding on the persistent fit's launch (option fit_fused, the default) against the
separate launches (fit_fused = 0, with vect_B by tiles so that the additions are the same).  argv[1] = workload
(small | cfg2 | headline), argv[2] = runs.  Prints ms per score, the sweep phase, the fit's launch time and whether the
result vectors and traces are identical."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth  # noqa: E402

W = {"small": (50_000, 525_000, 25, 128, 400), "cfg2": (100_000, 1_050_000, 50, 64, 400),
     "headline": (1_000_000, 10_500_000, 500, 128, 4000), "mid": (300_000, 3_150_000, 150, 128, 2000)}
name = sys.argv[1] if len(sys.argv) > 1 else "small"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, m, C, d, land = W[name]
g = synth.abcd_like(n, m, C, d, seed=42)
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
ctx.profile_enable(True)
out = {}
for label, opts in (("fused", {"fit_fused": 1}), ("separate+tiles", {"fit_fused": 0, "bvec_blocks": 1}),
                    ("separate+rows", {"fit_fused": 0, "bvec_blocks": 0})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    r = ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)
    tr = ctx.last_trace
    ctx.profile_reset()
    t0 = time.time()
    for _ in range(runs):
        r2 = ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)
    dt = (time.time() - t0) / runs * 1e3
    assert np.array_equal(r, r2)
    pr = ctx.profile()
    fit = pr.get("fit_persistent", {})
    ph = ctx.phase_ms()
    out[label] = (r, tr)
    print(f"{label:16s}: {dt:8.3f} ms per score; fit {fit.get('total_ms', 0) / max(1, fit.get('launches', 1)):.4f} ms x "
          f"{fit.get('launches', 0) // max(1, runs)} launches; fused alphas {ctx.get_stat('fit_fused_alphas')}; "
          f"sweep {ph.get('sweep', float('nan')):.3f} ms; result {[float(x) for x in r]}", flush=True)
    print("   iters", list(tr["iters"]), flush=True)
a, b, c2 = out["fused"], out["separate+tiles"], out["separate+rows"]
print("fused == separate+tiles (bits):", np.array_equal(a[0], b[0]), "traces:",
      np.array_equal(np.asarray(a[1]["div"]), np.asarray(b[1]["div"]), equal_nan=True),
      np.array_equal(np.asarray(a[1]["auc"]), np.asarray(b[1]["auc"]), equal_nan=True),
      list(a[1]["iters"]) == list(b[1]["iters"]))
print("fused vs separate+rows: max rel diff of the result", float(np.max(np.abs(a[0] - c2[0]) / np.maximum(1e-300, np.abs(c2[0])))),
      "iters equal:", list(a[1]["iters"]) == list(c2[1]["iters"]))
