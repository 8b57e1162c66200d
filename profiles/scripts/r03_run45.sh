set -o pipefail
OUT=gpurun_out/r4q; mkdir -p $OUT
CGE_FIT_FUSED_POW=1 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep" > $OUT/t1.log 2>&1; echo "fused=1 rc=$? $(tail -n 1 $OUT/t1.log)"
CGE_FIT_FUSED_POW=0 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep" > $OUT/t0.log 2>&1; echo "fused=0 rc=$? $(tail -n 1 $OUT/t0.log)"
