#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f
mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_wrappers_gpu.py -q -x -m gpu -k "one_autograd_node or new_dropout_masks" > $O/tests.txt 2>&1; tail -60 $O/tests.txt
