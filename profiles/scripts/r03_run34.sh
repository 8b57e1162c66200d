set -o pipefail
OUT=gpurun_out/r4f; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run b_all 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all
run b_plain 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run t_lm 700 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks or randomised_parity or split_global or wide"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4f/b_all.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("profile-all", round(j["ms_per_step"],2), {n: round(k[n]["total_ms_per_step"],3) for n in ("segmented_sort","children_sort","rss_rounds","sorted_prefix","group_project","group_eig","group_stats") if n in k}, "lm", round(j["phases_ms"]["landmarks"],2))
j=json.loads(open("gpurun_out/r4f/b_plain.log").read().strip().splitlines()[-1])
print("plain", round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), j["result"][:2])
PY
