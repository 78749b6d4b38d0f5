#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c6; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_dropout_gpu.py -q -m gpu -p no:cacheprovider -k "attention or dropout or blocks" -x > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v Warn $O/tests.log | tail -n 4
for m in chunked roles; do
  echo "== $m B=384"; VLNI_ATTN_BWD=$m B=384 timeout -k 10 300 python3 tools/attn_probe.py 2>&1 | grep -v amdgpu | head -5
done
echo "== roles B=64"; VLNI_ATTN_BWD=roles B=64 timeout -k 10 300 python3 tools/attn_probe.py 2>&1 | grep -v amdgpu | head -5
