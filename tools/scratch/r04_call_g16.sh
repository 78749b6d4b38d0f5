mkdir -p gpurun_out/r4g
for v in base new; do
  if [ $v = base ]; then export VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so; else unset VLNI_LIB_PATH; fi
  STEP_KINDS=1 M0=5504 M1=2752 GRAPH=1 NT_VARIANTS=5,14,15,32 NN_VARIANTS=5 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/ek_$v.log 2>&1
  STEP_KINDS=1 M0=33024 M1=16512 GRAPH=1 NT_VARIANTS=15,32 NN_VARIANTS=5 timeout -k 10 300 python tools/gemm_step_probe.py > gpurun_out/r4g/ekL_$v.log 2>&1
done
unset VLNI_LIB_PATH
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gemm or timed or epilogue" > gpurun_out/r4g/t16.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b16.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b16base.log 2>&1
