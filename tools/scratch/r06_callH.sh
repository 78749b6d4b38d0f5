#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h
mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_wrappers_gpu.py -q -x -m gpu > $O/tests.txt 2>&1; grep -v "^  File" $O/tests.txt | tail -25
for f in hamt duet; do for g in 0 1; do echo "== $f VLNI_GRAPHED_MODES=$g"; VLNI_GRAPHED_MODES=$g timeout -k 10 300 python3 tools/dropin_probe.py $f 2>&1 | grep -v amdgpu.ids | tail -3; done; done
