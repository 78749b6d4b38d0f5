mkdir -p gpurun_out/r4g
timeout -k 10 1150 python -m pytest tests -q -m gpu > gpurun_out/r4g/full21.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4g/full21.log
