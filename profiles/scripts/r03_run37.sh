set -o pipefail
OUT=gpurun_out/r4i; mkdir -p $OUT
CGE_REHEARSAL_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29527 bench.py --gpus 2 --steps 2 --warmup 1 --scale 0.25 > $OUT/b_n2.log 2> $OUT/b_n2.err
rc=$?; tail -n 3 $OUT/b_n2.err; [ $rc -eq 0 ] || exit 1
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4i/b_n2.log").read().strip().splitlines()[-1])
print(j["n_gpus"], round(j["ms_per_step"],1), j["ingest"][:80], j["result"][:2], j["collectives"]["backend"], j["independent_embeddings"]["embeddings_per_s"])
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/b_default.log 2> $OUT/b_default.err; echo "default rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4i/b_default.log").read().strip().splitlines()[-1])
print(round(j["ms_per_step"],2), j["value"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["kernels"]["edge_scatter"].get("back_to_back_frac"))
PY
