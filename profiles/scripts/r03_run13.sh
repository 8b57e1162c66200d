python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench
from cge.jl_amd import api, synth
wl = bench.WORKLOADS["headline"]
g = synth.abcd_like(wl["n"], int(wl["m"] * 1.05), wl["C"], wl["d"], seed=42)
ctx = api.Context(0)
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
for pct in (0, -1, 20, 60, 100):
    ctx.set_option("speculation_pct", pct)
    ctx.score(g["clusters"], wl["land"], 4, "rss", seed=42, auc_samples=10000)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.score(g["clusters"], wl["land"], 4, "rss", seed=42, auc_samples=10000)
    dt = (time.perf_counter() - t0) / 5
    ph = ctx.phase_ms()
    print(f"speculation_pct {pct:4d}: step {dt*1e3:6.2f} ms  landmarks {ph['landmarks']:6.2f}  batches {ctx.get_stat('landmark_batches')}  splits {ctx.get_stat('landmark_splits')}  rows {ctx.get_stat('landmark_batch_rows')}")
ctx.close()
PY
