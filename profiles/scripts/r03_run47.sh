set -o pipefail
OUT=gpurun_out/r4s; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_skew 300 python -m pytest tests/test_gpu_parity.py -q -x -k "start_skew"
for m in 0 1 0 1; do
  CGE_FIT_FUSED_POW=$m timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-back-to-back > $OUT/b_$m.log 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/b_$m.log").read().strip().splitlines()[-1])
print("fused_pow=$m", round(j["ms_per_step"],2), "sweep", round(j["phases_ms"]["sweep"],2), "fit", round(j["kernels"]["fit_persistent"]["avg_launch_ms"],4), j["result"][:2])
PY
done
run t_par 900 python -m pytest tests/test_gpu_parity.py -q -x -k "not landmarks_parity"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
run t_two 300 python -m pytest tests/test_gpu_two_ranks.py -q -x
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
