set -o pipefail
OUT=gpurun_out/r4d; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
for m in 0 1 0 1; do
  CGE_RSS2_LDS=$m timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline --profile-all > $OUT/b_$m.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$m.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("CGE_RSS2_LDS=$m", round(j["ms_per_step"],2), "rss2_walk", round(k["rss2_walk"]["total_ms_per_step"],3), "lm", round(j["phases_ms"]["landmarks"],2), j["result"][:2])
PY
done
run t_lm 700 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks or randomised_parity or split_global or wide"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
