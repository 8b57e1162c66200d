# offline stress: the randomised parity sweep with the wide embedding widths (the CPU oracle's Jacobi bounds the case rate)
set -o pipefail
OUT=gpurun_out/r4u; mkdir -p $OUT
CGE_STRESS_OFFSET=400 CGE_STRESS_WIDE=1 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity" > $OUT/stress_wide.log 2>&1; rc=$?
echo "wide stress rc=$rc: $(tail -n 1 $OUT/stress_wide.log)"; head -c 300 $OUT/stress_wide.log
