set -o pipefail
OUT=gpurun_out/r3k; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_dir 600 python -m pytest tests/test_gpu_parity.py -q -x -k "directed or ties or randomised_parity or handoffs"
run t_cfg4 600 python -m pytest tests/test_gpu_configs.py -q -x -k "config4"
run b_cfg4 300 python bench.py --workload cfg4 --steps 8 --warmup 3 --no-cpu-baseline
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python /dev/stdin <<'PY'
import json
txt=open('gpurun_out/r3k/b_cfg4.log').read()
b=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print('cfg4 step', b['ms_per_step'], 'fit', b['kernels'].get('fit_persistent'), {k:round(v,2) for k,v in b['phases_ms'].items() if k in('landmarks','diameter','sweep')})
PY
