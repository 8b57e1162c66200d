set -o pipefail
OUT=gpurun_out/r4r; mkdir -p $OUT
CGE_FIT_FUSED_POW=1 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -x -s -k "randomised_parity_sweep and 16" > $OUT/dbg.log 2>&1; echo "rc=$?"
grep "DBG" $OUT/dbg.log | head -12
