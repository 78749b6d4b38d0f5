#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_wrappers_gpu.py -q -x -m gpu -k "two_rollouts" > $O/t.txt 2>&1; grep -v "^  File" $O/t.txt | tail -25
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
