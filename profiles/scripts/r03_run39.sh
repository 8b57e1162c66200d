# offline stress: the randomised parity sweep under other seeds (CGE_STRESS_OFFSET) and the wide embedding widths
set -o pipefail
OUT=gpurun_out/r4k; mkdir -p $OUT
for off in 100 200 300; do
  CGE_STRESS_OFFSET=$off timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity" > $OUT/stress_$off.log 2>&1; rc=$?
  echo "offset $off rc=$rc: $(tail -n 1 $OUT/stress_$off.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
done
