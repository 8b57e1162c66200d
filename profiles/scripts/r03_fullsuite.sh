# the whole GPU suite, as the driver runs it
mkdir -p gpurun_out/r3s
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3s/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -n 6 gpurun_out/r3s/pytest_gpu.log
