# round 3, GPU call 4: two-lane runsplit -- parity tests, A/B bench on every workload
set -o pipefail
OUT=gpurun_out/r3d; mkdir -p $OUT
export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_par 600 python -m pytest tests/test_gpu_parity.py -q -x
for w in headline cfg2 cfg3 cfg4; do
  for l in 1 2; do
    CGE_LANES=$l run b_${w}_l$l 240 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --serial-diameter
  done
done
run b_headline_side 240 python bench.py --steps 8 --warmup 3 --no-cpu-baseline
run t_cfg 900 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5"
for f in $OUT/t_*.log; do echo "== $f"; tail -n 4 $f; done
