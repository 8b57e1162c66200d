"""Debug aid: config 3 (or argv[1]) with runsplit_lanes = 2, argv[2] times; counts the runs that fail or differ."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cge.jl_amd import api, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = bench.WORKLOADS[name]
g = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"], seed=42, directed=bool(wl.get("directed", False)))
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
ctx.set_option("diameter", 0)
ref = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], directed=bool(wl.get("directed", False)), seed=42, auc_samples=wl["samples"]).copy()
ctx.set_option("runsplit_lanes", 2)
bad = diff = 0
for rep in range(reps):
    try:
        r = ctx.score(g["clusters"], wl["land"], wl["forced"], wl["method"], directed=bool(wl.get("directed", False)), seed=42, auc_samples=wl["samples"])
        diff += int(not np.array_equal(r, ref))
    except Exception as e:
        bad += 1
print(f"{name} lanes=2: {reps} runs, {bad} errors, {diff} different results", flush=True)
