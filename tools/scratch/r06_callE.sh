#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e
mkdir -p $O; cd $R
T="timeout -k 10"
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
for cfg in "VLNI_TN_RING8=0 VLNI_GEMM_DESYNC=0" "VLNI_TN_RING8=1 VLNI_GEMM_DESYNC=0" "VLNI_TN_RING8=1 VLNI_GEMM_DESYNC=1"; do
env $cfg $T 400 python3 bench.py $A > $O/b.json 2> $O/b.err; echo "$cfg: $(python3 -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'], d['ms_per_step_median'])")"
done; done
