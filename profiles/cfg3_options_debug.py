"""Debug aid: config 3 under the alternative execution options of tests/test_gpu_configs.py, several times each."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cge.jl_amd import api, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = bench.WORKLOADS[name]
g = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"], seed=42, directed=bool(wl.get("directed", False)))
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
ctx.set_option("diameter", 0)
ref = None
seq = ({}, {"early_diameter": 1}, {"runsplit_lanes": 2}, {"speculation_pct": -1}, {"side_samples": 1}, {"cov_derive": 1})
if len(sys.argv) > 3: seq = ({}, {"early_diameter": 1}, {"runsplit_lanes": 2}) * int(sys.argv[3])
for opts in seq:
    for k, v in opts.items(): ctx.set_option(k, v)
    for rep in range(reps):
        try:
            r = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], directed=bool(wl.get("directed", False)), seed=42,
                          auc_samples=wl["samples"])
            if ref is None: ref = r.copy()
            print(opts, rep, "same" if np.array_equal(r, ref) else f"DIFFERENT {list(r)}", flush=True)
        except Exception as e:
            print(opts, rep, "ERROR", str(e)[:150], flush=True)
    for k in opts: ctx.set_option(k, {"runsplit_lanes": 1}.get(k, 0))
