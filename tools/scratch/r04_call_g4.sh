set -e
mkdir -p gpurun_out/r4g
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune.pkl > gpurun_out/r4g/ov1.log 2>&1
VLNI_OVERLAP_HISTORY=0 VLNI_GEMM_BREAKDOWN=14 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --load-tune gpurun_out/r4g/tune.pkl > gpurun_out/r4g/ov0.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune.pkl > gpurun_out/r4g/ov1b.log 2>&1
