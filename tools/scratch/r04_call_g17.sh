mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gemm or timed or epilogue or linear or ffn or block" > gpurun_out/r4g/t17.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline --dump-tune gpurun_out/r4g/tune17.pkl > gpurun_out/r4g/b17.log 2>&1
VLNI_LIB_PATH=$PWD/vln-imagine_amd/build/variants/libvlni_base.so timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > gpurun_out/r4g/b17base.log 2>&1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr17 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline --load-tune $GRAFT_REPO_ROOT/gpurun_out/r4g/tune17.pkl > $GRAFT_REPO_ROOT/gpurun_out/r4g/p17.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr17 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4g/stats17.csv
