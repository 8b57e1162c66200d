# quick validation: unique-row clamp on the device set, truncation path, configs
set -o pipefail
OUT=gpurun_out/r3i; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_trunc 300 python -m pytest tests/test_gpu_parity.py -q -x -k "truncation or duplicate or landmarks_parity_reference or diameter"
run t_cfg 600 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
run b_headline 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
python /dev/stdin <<'PY'
import json
txt=open('gpurun_out/r3i/b_headline.log').read()
b=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print('step', b['ms_per_step'], {k:round(v,2) for k,v in b['phases_ms'].items()})
PY
