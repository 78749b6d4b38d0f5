#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k
mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/aten_sites.py > $O/aten_hamt.txt 2>&1; tail -70 $O/aten_hamt.txt
