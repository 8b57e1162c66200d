"""Where the host-to-device time of the headline goes (round 5): the three cge_set_* calls timed one by one, five times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth  # noqa: E402

g = synth.abcd_like(1_000_000, 10_500_000, 500, 128, seed=42)
ctx = api.Context()
for rep in range(5):
    t0 = time.perf_counter()
    ctx.set_graph(g["edges"], g["eweights"], g["n"])
    t1 = time.perf_counter()
    ctx.set_vertex_data(g["comm"], g["vweights"])
    t2 = time.perf_counter()
    ctx.set_embedding(g["embedding"])
    t3 = time.perf_counter()
    print(f"set_graph {1e3 * (t1 - t0):.2f} ms  set_vertex_data {1e3 * (t2 - t1):.2f} ms  set_embedding {1e3 * (t3 - t2):.2f} ms  "
          f"total {1e3 * (t3 - t0):.2f} ms", flush=True)
r = ctx.score(g["clusters"], 4000, 4, "rss", seed=42, auc_samples=10000)
print([float(x) for x in r])
