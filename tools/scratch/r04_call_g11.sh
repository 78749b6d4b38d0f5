mkdir -p gpurun_out/r4g
T="tests/test_fulldepth_gpu.py -q -m gpu -s -k hamt-64-low2"
VLNI_GELU_STORE_GRAD=0 timeout -k 10 280 python -m pytest $T > gpurun_out/r4g/f_gelu0.log 2>&1
VLNI_LANG_QKV_ONCE=0 timeout -k 10 280 python -m pytest $T > gpurun_out/r4g/f_qkv0.log 2>&1
VLNI_ATTN_BWD=chunked timeout -k 10 280 python -m pytest $T > gpurun_out/r4g/f_chunk.log 2>&1
true
