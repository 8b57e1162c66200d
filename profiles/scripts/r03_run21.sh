set -o pipefail
OUT=gpurun_out/r3s; mkdir -p $OUT
for T in 256 512 900; do for f in 0 5; do
  CGE_EIG_FORM=$f timeout -k 10 120 python profiles/eig_stage_timing.py $T 128 >> $OUT/eig.txt 2>&1 || exit 1
done; done
for f in 0 5; do CGE_EIG_FORM=$f timeout -k 10 120 python profiles/eig_stage_timing.py 512 64 >> $OUT/eig.txt 2>&1 || exit 1; done
for f in 0 5; do CGE_EIG_FORM=$f timeout -k 10 120 python profiles/eig_stage_timing.py 512 32 >> $OUT/eig.txt 2>&1 || exit 1; done
cat $OUT/eig.txt
CGE_EIG_FORM=5 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "eig or landmarks or wide" > $OUT/t_eig5.log 2>&1; echo "t_eig5 rc=$?"; tail -n 3 $OUT/t_eig5.log
CGE_EIG_FORM=5 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b5.log 2>&1; echo "b5 rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b0.log 2>&1; echo "b0 rc=$?"
python - <<'PY'
import json
for w in ("b0","b5"):
    j=json.loads(open(f"gpurun_out/r3s/{w}.log").read().strip().splitlines()[-1])
    print(w, round(j["ms_per_step"],2), "lm", round(j["phases_ms"]["landmarks"],2), "eig", round(j["kernels"]["group_eig"]["total_ms_per_step"],3), j["result"][:2])
PY
