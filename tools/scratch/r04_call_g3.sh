set -e
mkdir -p gpurun_out/r4g
echo "== old" > gpurun_out/r4g/ln.log
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_lnold.so timeout -k 10 200 python tools/ln_probe.py >> gpurun_out/r4g/ln.log 2>&1
echo "== new cap 128" >> gpurun_out/r4g/ln.log
timeout -k 10 200 python tools/ln_probe.py >> gpurun_out/r4g/ln.log 2>&1
echo "== new cap 256" >> gpurun_out/r4g/ln.log
VLNI_LN_BWD_BLOCKS=256 timeout -k 10 200 python tools/ln_probe.py >> gpurun_out/r4g/ln.log 2>&1
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layer or norm or ln" >> gpurun_out/r4g/ln.log 2>&1
