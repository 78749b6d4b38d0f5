#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py -q -x -k "uploaded_inside" > $O/t.log 2>&1; tail -3 $O/t.log; grep -n "^E  " $O/t.log | cut -c1-300 | head
