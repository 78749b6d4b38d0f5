mkdir -p gpurun_out/r4g
T="tests/test_fulldepth_gpu.py -q -m gpu -s -k hamt-64-low2"
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_oldattn.so timeout -k 10 280 python -m pytest $T > gpurun_out/r4g/f_oldattn.log 2>&1
true
