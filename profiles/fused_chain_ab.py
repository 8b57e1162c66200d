"""A/B aid (round 5): the alpha chain riding on the persistent fit's launch (option fit_fused, the default) against the
separate launches (fit_fused = 0), for the strip form of the fit (kernels_fits.hip, the default) and the tile form (fit_strip = 0).  argv[1] = workload
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
for label, opts in (("strip fused", {"fit_strip": 1, "fit_fused": 1}), ("tile fused", {"fit_strip": 0, "fit_fused": 1}),
                    ("strip separate", {"fit_strip": 1, "fit_fused": 0, "bvec_blocks": 1}),
                    ("tile separate", {"fit_strip": 0, "fit_fused": 0, "bvec_blocks": 1}),
                    ("separate+rows", {"fit_strip": 1, "fit_fused": 0, "bvec_blocks": 0})):
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
ref = out["separate+rows"]
for label, (r, tr) in out.items():
    print(f"{label:16s} vs separate+rows: max rel diff of the result {float(np.max(np.abs(r - ref[0]) / np.maximum(1e-300, np.abs(ref[0])))):.3e}; "
          f"iters equal {list(tr['iters']) == list(ref[1]['iters'])}; "
          f"max rel diff of the div trace {float(np.nanmax(np.abs(np.asarray(tr['div']) - np.asarray(ref[1]['div'])) / np.maximum(1e-300, np.abs(np.asarray(ref[1]['div']))))):.3e}; "
          f"auc trace {float(np.nanmax(np.abs(np.asarray(tr['auc']) - np.asarray(ref[1]['auc'])))):.3e}")
