#!/bin/bash
# sharded ingest: two ranks on one GPU (hook path) + the bench's N = 2 rehearsal
set -o pipefail
mkdir -p gpurun_out/r3o
timeout -k 10 800 python -m pytest tests/test_gpu_two_ranks.py -x -q > gpurun_out/r3o/t_two.log 2>&1
rc=$?; tail -n 15 gpurun_out/r3o/t_two.log; [ $rc -eq 0 ] || exit 1
CGE_REHEARSAL_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --scale 0.25 > gpurun_out/r3o/b_n2.log 2> gpurun_out/r3o/b_n2.err
rc=$?; tail -n 5 gpurun_out/r3o/b_n2.err; tail -c 1500 gpurun_out/r3o/b_n2.log; [ $rc -eq 0 ] || exit 1
CGE_SHARD_INGEST=0 CGE_REHEARSAL_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --scale 0.25 > gpurun_out/r3o/b_n2_full.log 2> gpurun_out/r3o/b_n2_full.err
rc=$?; tail -c 1500 gpurun_out/r3o/b_n2_full.log; [ $rc -eq 0 ] || exit 1
