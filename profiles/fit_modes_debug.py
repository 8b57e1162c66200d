"""Debug aid: the score of one graph under the fit modes 1 (launch per iteration) and 2 (persistent), with the iteration counts."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cge.jl_amd import api, synth
ctx = api.Context()
g = synth.abcd_like(60000, 600000, 40, 16, seed=5)
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
for mode in (1, 2, 1, 2):
    ctx.set_option("fit_persistent", mode)
    r = ctx.score(g["clusters"], 4000, 4, "rss", seed=3, auc_samples=5000)
    print(mode, [float(x) for x in r], "iters", ctx.get_stat("fit_iterations"), "fallbacks", ctx.get_stat("fit_persistent_fallbacks"),
          "landmarks", ctx.get_stat("landmarks"), "splits", ctx.get_stat("landmark_splits"))
