set -o pipefail
OUT=gpurun_out/r4g; mkdir -p $OUT
for q in 0 1 2 0 1 2; do
  CGE_SPEC_QUANT=$q timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-back-to-back > $OUT/b_$q.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$q.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("quant=$q", round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), "eig launches/step", k["group_eig"]["launches"]/10, "eig ms", round(k["group_eig"]["total_ms_per_step"],2), "cov", round(k["group_stats"]["total_ms_per_step"],2))
PY
done
