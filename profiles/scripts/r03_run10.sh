set -o pipefail
OUT=gpurun_out/r3j; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "edge_scatter" > $OUT/t_scatter.log 2>&1; echo "rc=$?"; tail -n 5 $OUT/t_scatter.log
