#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
timeout -k 10 500 python tools/small_gemm_probe.py duet > $O/duet.log 2>&1; tail -40 $O/duet.log
