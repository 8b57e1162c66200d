set -o pipefail
OUT=gpurun_out/r3m; mkdir -p $OUT
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_lm 600 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks or randomised_parity or split_global or wide"
run t_cfg 700 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for w in headline cfg2 cfg3; do run b_$w 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline; done
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
