set -o pipefail
OUT=gpurun_out/r4m; mkdir -p $OUT
CGE_FIT_FUSED_POW=0 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -x -k "randomised_parity_sweep and (1 or 2 or 3)" > $OUT/t0.log 2>&1; echo "fused=0 rc=$? $(tail -n 1 $OUT/t0.log)"
CGE_FIT_FUSED_POW=1 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep" > $OUT/t1.log 2>&1; echo "fused=1 rc=$? $(tail -n 1 $OUT/t1.log)"
grep -n "^FAILED" $OUT/t1.log | head -40
