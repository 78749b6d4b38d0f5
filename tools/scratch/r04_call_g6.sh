mkdir -p gpurun_out/r4g
VLNI_EPISODE_MASKS=0 timeout -k 10 300 python -m pytest tests/test_tape_gpu.py -q -m gpu -k full_width > gpurun_out/r4g/t6a.log 2>&1
timeout -k 10 300 python -m pytest tests/test_tape_gpu.py -q -m gpu -k full_width > gpurun_out/r4g/t6b.log 2>&1
