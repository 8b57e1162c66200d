"""Diagnostics (round 5): runs the headline sweep on a library built with -DCGE_FLOW_CLOCK (csrc/build/libcge_hip_clock.so:
`hipcc ... -DCGE_FLOW_CLOCK -c kernels_fitp.hip` linked with the other objects), whose fused persistent fit prints wall-clock
stamps of its sections (prologue / loop / vect_B epilogue / tallies) from two waves per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
api._LIB_PATH = os.path.join(os.path.dirname(api._LIB_PATH), sys.argv[2] if len(sys.argv) > 2 else "libcge_hip_clock.so")
W = {"small": (50_000, 525_000, 25, 128, 400), "headline": (1_000_000, 10_500_000, 500, 128, 4000)}
n, m, C, d, land = W[name]
g = synth.abcd_like(n, m, C, d, seed=42)
ctx = api.Context()
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
for _ in range(2):
    print("score", [float(x) for x in ctx.score(g["clusters"], land, 4, "rss", seed=42, auc_samples=10000)], flush=True)
