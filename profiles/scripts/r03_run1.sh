set -o pipefail
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/r3a/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return 0
}
run t_small 420 python -m pytest tests/test_gpu_parity.py -q -x -k "ties or edge_scatter or landmarks_parity_reference or readme or wgcl_reference" 
run t_two 300 python -m pytest tests/test_gpu_two_ranks.py -q -x -k "abandoned or other_split"
run b_head 240 python bench.py --steps 10 --warmup 3 --profile-all
run b_serial 240 python bench.py --steps 10 --warmup 3 --serial-diameter --no-cpu-baseline
run t_cfg 1000 python -m pytest tests/test_gpu_configs.py -q -x
tail -3 gpurun_out/r3a/t_small.log gpurun_out/r3a/t_two.log gpurun_out/r3a/t_cfg.log
