#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c15; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "resident or attention" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v Warn $O/tests.log | grep "^FAILED\|^E  \|passed\|failed" | cut -c1-300 | head -20
