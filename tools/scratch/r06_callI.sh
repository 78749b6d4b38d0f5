#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i
mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests/test_hamt_gpu.py tests/test_duet_gpu.py -q -x -m gpu -k "dropin" > $O/tests.txt 2>&1; grep -v "^  File" $O/tests.txt | tail -40
