set -o pipefail
OUT=gpurun_out/r4h; mkdir -p $OUT
for pct in 10 15 20 26; do
  CGE_SPEC_PCT=$pct timeout -k 10 300 python bench.py --workload cfg3 --steps 8 --warmup 3 --no-cpu-baseline --no-back-to-back > $OUT/b_$pct.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$pct.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("cfg3 pct=$pct", round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), "eig launches/step", k["group_eig"]["launches"]/8, "eig ms", round(k["group_eig"]["total_ms_per_step"],2), "cov", round(k["group_stats"]["total_ms_per_step"],2))
PY
done
