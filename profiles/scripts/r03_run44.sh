set -o pipefail
OUT=gpurun_out/r4p; mkdir -p $OUT
CGE_FIT_FUSED_POW=0 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep and (16 or 20 or 24 or 1)" > $OUT/a.log 2>&1; echo "unfused nodelayenv rc=$? $(tail -n 1 $OUT/a.log)"
grep "^FAILED" $OUT/a.log | head
CGE_FIT_FUSED_POW=0 CGE_FIT_DELAY=0 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep and (16 or 20 or 24 or 1)" > $OUT/b.log 2>&1; echo "unfused delay0 rc=$? $(tail -n 1 $OUT/b.log)"
grep "^FAILED" $OUT/b.log | head
