"""A/B aid: the rss2 rule at d = 128 (config 2's graph with a 128-wide embedding): argv[1] = runs.  Prints ms per score and the result."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth
g = synth.abcd_like(100000, 1050000, 50, 128, seed=42)
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
ctx.profile_enable(True)
r = ctx.score(g["clusters"], 400, 4, "rss2", seed=42, auc_samples=10000)
ctx.profile_reset()
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n): r = ctx.score(g["clusters"], 400, 4, "rss2", seed=42, auc_samples=10000)
dt = (time.time() - t0) / n * 1e3
pr = ctx.profile().get("rss2_walk", {})
print(f"CGE_RSS2_LDS2={os.environ.get('CGE_RSS2_LDS2','default')}: {dt:.2f} ms per score (timers on), rss2_walk {pr.get('total_ms',0)/max(1,pr.get('launches',1)):.4f} ms per launch, result {[float(x) for x in r]}")
