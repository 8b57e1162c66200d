"""`--force-exact` (v_to_l = Int[], N = n) on the device at sizes far beyond the reference's 10 000-vertex switch to
landmarks (src/auxilary.jl:194-197): `python3 profiles/exact_mode_probe.py n [d]` prints time, iterations, result."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cge.jl_amd import api, synth  # noqa: E402

n = int(sys.argv[1])
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = synth.abcd_like(n, 10 * n, max(8, n // 750), d, seed=7)
ctx = api.Context(0)
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
if os.environ.get("PROBE_KERNEL_TIMERS"):  # per-kernel-family event timers (a few us of serialisation each)
    ctx.profile_select(())
    ctx.profile_enable(True)
    ctx.profile_reset()
t0 = time.perf_counter()
res = ctx.score([], -1, seed=3, auc_samples=10000)
t = time.perf_counter() - t0
if os.environ.get("PROBE_KERNEL_TIMERS"):
    for k, v in sorted(ctx.profile().items(), key=lambda kv: -kv[1]["total_ms"])[:12]:
        print(f"  {k:20s} launches {v['launches']:6d} total {v['total_ms']:10.1f} ms", flush=True)
print(f"n={n} m={g['m']} d={d}: {t:.2f} s, {ctx.get_stat('fit_iterations')} Chung-Lu iterations over {ctx.last_trace['n_alpha']} "
      f"alphas, persistent alphas {ctx.get_stat('fit_persistent_alphas')}, result {[float(x) for x in res]}", flush=True)
print({k: round(v, 1) for k, v in ctx.phase_ms().items() if v > 1})
ctx.close()
