#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_wrappers_gpu.py -q -x -m gpu > $O/t.txt 2>&1; grep -v "^  File" $O/t.txt | tail -4
for f in hamt duet; do timeout -k 10 300 python3 tools/dropin_probe.py $f 2>&1 | grep -v amdgpu.ids | tail -2; done
