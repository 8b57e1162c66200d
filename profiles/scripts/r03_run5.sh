# round 3, GPU call 5: rehearsal-based speculation (default) vs the rank rule of rounds 1-2 (CGE_SPEC_PCT), all workloads
set -o pipefail
OUT=gpurun_out/r3e; mkdir -p $OUT
export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_quick 300 python -m pytest tests/test_gpu_parity.py -q -x -k "landmarks_parity or randomised_parity"
for w in headline cfg2 cfg3 cfg4; do
  run b_${w}_new 240 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline
  p=40; if [ $w = cfg3 ]; then p=10; fi
  CGE_SPEC_PCT=$p run b_${w}_old 240 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline
done
run b_headline_noside 240 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side
run b_cfg5_new 500 python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline
for f in $OUT/t_*.log; do echo "== $f"; tail -n 4 $f; done
