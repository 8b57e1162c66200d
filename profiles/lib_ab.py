"""A/B aid: one workload on an alternative build of the library.  argv: <library file under csrc/build> <workload> [runs].
Prints ms per score and the kernel timers (profile on)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth  # noqa: E402

api._LIB_PATH = os.path.join(os.path.dirname(api._LIB_PATH), sys.argv[1])
W = {"cfg5_200k": (200_000, 4_200_000, 1500, 512, 12000, "rss"), "headline": (1_000_000, 10_500_000, 500, 128, 4000, "rss"),
     "cfg2": (100_000, 1_050_000, 50, 64, 400, "rss2"), "cfg3": (1_000_000, 21_000_000, 500, 128, 4000, "diameter")}
n, m, C, d, land, method = W[sys.argv[2]]
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = synth.abcd_like(n, m, C, d, seed=42)
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
r = ctx.score(g["clusters"], land, 4, method, seed=42, auc_samples=10000)
ctx.profile_enable(True)
ctx.score(g["clusters"], land, 4, method, seed=42, auc_samples=10000)
ctx.profile_reset()
t0 = time.time()
for _ in range(runs):
    r = ctx.score(g["clusters"], land, 4, method, seed=42, auc_samples=10000)
dt = (time.time() - t0) / runs * 1e3
print(f"{sys.argv[1]} {sys.argv[2]}: {dt:.2f} ms per score (timers on); result {[float(x) for x in r]}")
for k, v in sorted(ctx.profile().items(), key=lambda kv: -kv[1].get("total_ms", 0))[:14]:
    print(f"   {k:24s} {v.get('total_ms', 0) / runs:9.3f} ms per score, {v.get('launches', 0) // runs} launches")
print("   phases", {k: round(v, 2) for k, v in ctx.phase_ms().items() if v > 1.0})
print("   batches", ctx.get_stat("landmark_batches"), "rows", ctx.get_stat("landmark_batch_rows"), "splits", ctx.get_stat("landmark_splits"))
