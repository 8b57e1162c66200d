set -o pipefail
OUT=gpurun_out/r4o; mkdir -p $OUT
for dly in 0 3 10; do
CGE_FIT_FUSED_POW=0 CGE_FIT_DELAY=$dly timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity_sweep and (16 or 20 or 24 or 1)" > $OUT/d_$dly.log 2>&1; echo "unfused delay=$dly rc=$? $(tail -n 1 $OUT/d_$dly.log)"
done
