# round 3, GPU call 6: bf16-split bound pass -- diameter tests, bench A/B (CGE default vs fp32 bound pass)
set -o pipefail
OUT=gpurun_out/r3f; mkdir -p $OUT
export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_diam 400 python -m pytest tests/test_gpu_parity.py -q -x -k "diameter or max_pair"
run b_headline 240 python bench.py --steps 10 --warmup 3 --profile-all
run b_cfg2 240 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline
run t_cfg 900 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for f in $OUT/t_*.log; do echo "== $f"; tail -n 4 $f; done
