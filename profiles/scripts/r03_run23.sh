set -o pipefail
OUT=gpurun_out/r3u; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or eig" > $OUT/t_wide.log 2>&1; echo "t_wide rc=$?"; tail -n 3 $OUT/t_wide.log
timeout -k 10 600 python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_cfg5.log 2> $OUT/b_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r3u/b_cfg5.log").read().strip().splitlines()[-1])
k=j["kernels"]
print("cfg5", round(j["ms_per_step"],1), {n: round(k[n]["total_ms_per_step"],1) for n in k}, {a: round(b,1) for a,b in j["phases_ms"].items() if a in ("landmarks","diameter","sweep","aggregate")})
PY
