set -o pipefail
OUT=gpurun_out/r3w; mkdir -p $OUT
for rep in 1 2 3; do for m in 1 0; do
  CGE_FLOW_INLINE_ARM=$m timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b_${m}_$rep.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_${m}_$rep.log").read().strip().splitlines()[-1])
print("inline_arm=$m rep=$rep", round(j["ms_per_step"],3), "sweep", round(j["phases_ms"]["sweep"],3))
PY
done; done
