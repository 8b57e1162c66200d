set -o pipefail
OUT=gpurun_out/r3y; mkdir -p $OUT
for w in cfg2 cfg3; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --profile-all > $OUT/b_$w.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$w.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("$w", round(j["ms_per_step"],2), {a: round(b,2) for a,b in j["phases_ms"].items() if not a.startswith("dm_")})
for n in sorted(k, key=lambda n:-k[n]["total_ms_per_step"])[:16]: print("   ", n, round(k[n]["total_ms_per_step"],3), k[n]["launches"], round(k[n]["avg_launch_ms"],4))
PY
done
