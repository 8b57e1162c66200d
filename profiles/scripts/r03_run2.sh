# round 3, GPU call 2: eigen-solver forms A/B (tests + bench), cfg5 full-size test, kernel trace of the headline
set -o pipefail
OUT=gpurun_out/r3b; mkdir -p $OUT
export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0; }
run t_eig4 300 python -m pytest tests/test_gpu_parity.py -q -x -k "group_eig or landmarks_parity or wide_embeddings or duplicate"
CGE_EIG_FORM=8 run t_eig8 300 python -m pytest tests/test_gpu_parity.py -q -x -k "group_eig or landmarks_parity_reference"
for f in 0 4 8; do
  CGE_EIG_FORM=$f run b_head_f$f 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-diameter
done
CGE_EIG_FORM=4 run b_head_side 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
for f in 0 4 8; do
  CGE_EIG_FORM=$f run b_cfg3_f$f 200 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --serial-diameter
done
run t_cfg5 600 python -m pytest tests/test_gpu_configs.py -q -x -k "config5_full"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --serial-diameter > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
for f in $OUT/t_*.log; do echo "== $f"; tail -n 3 $f; done
