#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; V=$R/vln-imagine_amd/build/variants
mkdir -p $O; cd $R
T="timeout -k 10"
for d in 0 1 4; do VLNI_GEMM_DESYNC=$d VLNI_LIB_PATH=$V/lib_S2.so $T 200 python3 tools/p8_stamps.py d$d > $O/p8_d$d.txt 2>&1; grep -v amdgpu $O/p8_d$d.txt; done
