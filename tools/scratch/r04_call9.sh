#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c9; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "ops tests rc=$?"; grep -v Warn $O/tests.log | grep "^FAILED\|^E  \|passed\|failed" | cut -c1-250 | head -30
timeout -k 10 900 python3 -m pytest tests/test_fulldepth_gpu.py -q -m gpu -p no:cacheprovider -s -k "timed_path" > $O/full.log 2>&1
echo "fulldepth rc=$?"; grep "vs fp32 HIP\|passed\|failed" $O/full.log | cut -c1-330
