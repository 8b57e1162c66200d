set -o pipefail
OUT=gpurun_out/r4b; mkdir -p $OUT
for pct in 40 50 60 75 100; do
  CGE_SPEC_PCT=$pct timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b_$pct.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$pct.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("pct=$pct", round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), "eig launches/step", k["group_eig"]["launches"]/10, "eig ms", round(k["group_eig"]["total_ms_per_step"],2), "cov", round(k["group_stats"]["total_ms_per_step"],2))
PY
done
