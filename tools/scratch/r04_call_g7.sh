mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_duet_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > gpurun_out/r4g/t7.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr7 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r4g/b7.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr7 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4g/stats7.csv
