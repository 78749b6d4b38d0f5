#!/bin/bash
# round 4, call 2: the stale-mask anomaly in its original habitat - the pytest process of tests/test_buckets_gpu.py - with the mask computed in-graph
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2
mkdir -p $O
VLNI_MASK_IN_GRAPH=1 timeout -k 10 600 python3 -m pytest tests/test_buckets_gpu.py -q -m gpu -p no:cacheprovider > $O/buckets_ingraph.log 2>&1
echo "buckets in-graph rc=$?"; tail -n 15 $O/buckets_ingraph.log
VLNI_MASK_IN_GRAPH=1 timeout -k 10 300 python3 -m pytest tests/test_buckets_gpu.py -q -m gpu -p no:cacheprovider -k stepped_inference > $O/inf_only.log 2>&1
echo "inference-only rc=$?"; tail -n 5 $O/inf_only.log
