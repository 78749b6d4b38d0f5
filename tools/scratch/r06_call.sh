#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_wrappers_gpu.py -q -x -m gpu > $O/t.txt 2>&1; grep -v "^  File" $O/t.txt | tail -30
