set -o pipefail
OUT=gpurun_out/r3v; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_par 900 python -m pytest tests/test_gpu_parity.py -q -x -k "not landmarks_parity and not randomised"
run t_two 600 python -m pytest tests/test_gpu_two_ranks.py -q -x
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for w in headline cfg2 cfg4; do run b_$w 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline; done
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python - <<'PY'
import json
for w in ("headline","cfg2","cfg4"):
    try:
        j=json.loads(open(f"gpurun_out/r3v/b_{w}.log").read().strip().splitlines()[-1])
        p=j["phases_ms"]
        print(w, round(j["ms_per_step"],2), {k: round(p[k],2) for k in ("landmarks","diameter","sweep") if k in p}, round(j["kernels"]["fit_persistent"]["avg_launch_ms"],4))
    except Exception as e: print(w, e)
PY
