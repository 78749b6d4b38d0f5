mkdir -p gpurun_out/r4g
timeout -k 10 900 python -m pytest tests/test_tape_gpu.py tests/test_buckets_gpu.py tests/test_hamt_gpu.py -q -m gpu > gpurun_out/r4g/t5.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b5.log 2>&1
