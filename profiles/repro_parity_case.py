"""Re-run one case of tests/test_gpu_parity.py::test_randomised_parity_sweep (with the wide dimensions of CGE_STRESS_WIDE)
through every fit form and the oracle, printing the per-alpha local-score tallies: python3 profiles/repro_parity_case.py <case>"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import cge.jl_amd as cg
from cge.jl_amd import api, synth
import importlib.util
spec = importlib.util.spec_from_file_location("cge_oracle_py", "/root/repo/oracle/oracle.py")
orc_mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(orc_mod)
case = int(sys.argv[1]) if len(sys.argv) > 1 else 212
rng = np.random.default_rng(1000 + case)
n = int(rng.integers(60, 1500))
dims = [2, 3, 5, 8, 17, 33, 64, 100] + [129, 130, 160, 192, 257, 300]
d = int(rng.choice(dims))
C = int(rng.integers(2, max(3, n // 25)))
method = ["rss", "rss2", "size", "diameter"][case % 4]
directed = bool(rng.integers(0, 2)); split = bool(rng.integers(0, 2)); forced = int(rng.choice([1, 2, 4]))
g = synth.abcd_like(n, int(rng.integers(3, 9)) * n, C, d, seed=500 + case, directed=directed)
land = int(min(n // 3, max(C * forced, rng.integers(C, 6 * C + 2))))
ew, vw = g["eweights"], g["vweights"]
if rng.integers(0, 2):
    ew = rng.integers(1, 17, size=len(ew)) / 4.0
    vw = np.zeros(n); np.add.at(vw, g["edges"][:, 0] - 1, ew); np.add.at(vw, g["edges"][:, 1] - 1, ew)
print("case", case, "n", n, "d", d, "C", C, method, "directed", directed, "split", split, "forced", forced, "land", land)
ctx = api.Context(0)
args = (g["edges"], ew, vw, g["clusters"], g["comm"], g["embedding"], False, land, forced, method, directed)
got = cg.landmarks(*args, ctx=ctx)
dii, lemb, lcomm, ledges, lw, lweight, v2l = got
S = 1500
if directed:
    p1, ni, nj = api.draw_samples(ctx, case, S, directed=True); smp, fn = (p1, ni, nj, p1), cg.wGCL_directed
else:
    smp, fn = api.draw_samples(ctx, case, S), cg.wGCL
wargs = (ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, g["edges"], ew, g["embedding"], split)
for opt in (2, 1, 3, 4):
    ctx.set_option("fit_persistent", opt)
    try:
        res, tr = fn(*wargs, case, S, samples=smp, trace=True, ctx=ctx)
    except Exception as e:
        print("opt", opt, "error", e); continue
    print("fit_persistent", opt, [float(x) for x in res], "auc trace", [round(a * S) for a in tr["auc"][:24]])
ctx.set_option("fit_persistent", 0)
orc = orc_mod
ofn = orc.wGCL_directed if directed else orc.wGCL
exp, etr = ofn(*wargs, smp, trace=True)
print("oracle        ", [float(x) for x in exp], "auc trace", [round(a * S) for a in etr["auc"][:24]])
ctx.close()
