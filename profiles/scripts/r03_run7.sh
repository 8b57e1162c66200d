# round 3, GPU call 7: bf16-split bound pass, second form (8 waves, batched loads)
set -o pipefail
OUT=gpurun_out/r3g; mkdir -p $OUT
export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_diam 400 python -m pytest tests/test_gpu_parity.py -q -x -k "diameter or max_pair"
run b_headline 240 python bench.py --steps 10 --warmup 3 --profile-all --no-cpu-baseline
for f in $OUT/t_*.log; do echo "== $f"; tail -n 4 $f; done
python - <<'PY'
import json
txt=open('gpurun_out/r3g/b_headline.log').read()
b=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print('step', b['ms_per_step'], 'pcent', b['kernels']['pcent'], 'diam', b['phases_ms']['diameter'], b['diameter'])
PY
