mkdir -p gpurun_out/r4g
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_dropout_gpu.py tests/test_hamt_gpu.py tests/test_duet_gpu.py -q -m gpu -x > gpurun_out/r4g/t14.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b14.log 2>&1
