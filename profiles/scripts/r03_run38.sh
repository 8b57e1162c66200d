set -o pipefail
OUT=gpurun_out/r4j; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
for w in headline cfg3 cfg2; do run b_$w 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-back-to-back; done
run t_lm 700 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks or randomised_parity or split_global or wide"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
run t_two 300 python -m pytest tests/test_gpu_two_ranks.py -q -x
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python - <<'PY'
import json
for w in ("headline","cfg3","cfg2"):
    j=json.loads(open(f"gpurun_out/r4j/b_{w}.log").read().strip().splitlines()[-1])
    print(w, round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2))
PY
