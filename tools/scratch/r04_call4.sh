#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hamt_gpu.py tests/test_duet_gpu.py tests/test_tape_gpu.py tests/test_buckets_gpu.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "rc=$?"; grep -v Warning $O/tests.log | tail -n 30
