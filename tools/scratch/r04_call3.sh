#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3
mkdir -p $O
for m in 1 2; do
VLNI_MASK_IN_GRAPH=$m timeout -k 10 600 python3 -m pytest tests/test_buckets_gpu.py tests/test_tape_gpu.py -q -m gpu -p no:cacheprovider -x > $O/m$m.log 2>&1
echo "mode $m rc=$?"; tail -n 6 $O/m$m.log
done
