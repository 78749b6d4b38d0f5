mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gemm or timed or epilogue or linear or ffn or block" > gpurun_out/r4g/t23.log 2>&1 || exit 1
for m in hamt; do
timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune23_$m.pkl > gpurun_out/r4g/b23_${m}_new.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune23_$m.pkl > gpurun_out/r4g/b23_${m}_base.log 2>&1
timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune23_$m.pkl > gpurun_out/r4g/b23_${m}_new2.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune gpurun_out/r4g/tune23_$m.pkl > gpurun_out/r4g/b23_${m}_base2.log 2>&1
done
