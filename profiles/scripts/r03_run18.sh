set -o pipefail
OUT=gpurun_out/r3q; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_tree 200 python -m pytest tests/test_gpu_parity.py -q -x -k "projection_reduction"
run t_lm 600 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks or randomised_parity or split_global or wide"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for w in headline cfg2 cfg3; do run b_$w 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline; done
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python - <<'PY'
import json
for w in ("headline","cfg2","cfg3"):
    try:
        j=json.loads(open(f"gpurun_out/r3q/b_{w}.log").read().strip().splitlines()[-1])
        k=j["kernels"]
        print(w, round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), {n: round(k[n]["avg_launch_ms"],4) for n in ("group_project","sorted_prefix","group_stats","group_eig") if n in k})
    except Exception as e: print(w, e)
PY
