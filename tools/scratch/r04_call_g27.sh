mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "timed_gemm" > gpurun_out/r4g/t27.log 2>&1 || exit 1
STEP_KINDS=1 M0=5504 M1=2752 GRAPH=1 NT_VARIANTS=14,15,32,33 NN_VARIANTS=5 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/ek27.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b27a.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b27b.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b27c.log 2>&1
