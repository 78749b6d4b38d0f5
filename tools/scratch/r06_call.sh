#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "128 1" "64 1" "32 1" "128 2" "128 4" "256 1"; do set -- $cfg; echo "== blocks $1 rows/wave $2"; VLNI_LN_BWD_BLOCKS=$1 VLNI_LN_BWD_ROWS=$2 timeout -k 10 120 python3 tools/ln_probe.py 2>&1 | grep -v amdgpu | head -5 | cut -c1-150; done
