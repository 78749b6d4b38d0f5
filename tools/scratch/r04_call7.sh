#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c7; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_dropout_gpu.py -q -m gpu -p no:cacheprovider -k "attention or dropout or blocks" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v Warn $O/tests.log | grep "^FAILED\|^E  \|passed\|failed" | cut -c1-200 | head
echo "== product B=384"; B=384 timeout -k 10 300 python3 tools/attn_probe.py 2>&1 | grep -v amdgpu | head -5
VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/st.so timeout -k 10 300 python3 tools/attn_stamps.py 0.1 2>&1 | grep -v "amdgpu\|alive" | head -10
