#!/bin/bash
# round 6, call B: ring kernel with the two wave rows one barrier apart (lib_E), start-time stagger of the persistent NT kernels (VLNI_GEMM_DESYNC)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; V=$R/vln-imagine_amd/build/variants
mkdir -p $O; cd $R
T="timeout -k 10"
$T 300 python3 tools/ring_probe.py base > $O/ring_base.txt 2>&1; tail -2 $O/ring_base.txt
VLNI_LIB_PATH=$V/lib_E.so $T 200 python3 tools/ring_probe.py E > $O/ring_E.txt 2>&1; tail -12 $O/ring_E.txt
for d in 0 1; do
VLNI_GEMM_DESYNC=$d T=6 GRAPH=1 STEP_KINDS=1 NT_VARIANTS=15,32 NN_VARIANTS=6 $T 300 python3 tools/gemm_step_probe.py > $O/step6_d$d.txt 2>&1; tail -12 $O/step6_d$d.txt
VLNI_GEMM_DESYNC=$d T=1 GRAPH=1 STEP_KINDS=1 NT_VARIANTS=15,32 NN_VARIANTS=6 $T 300 python3 tools/gemm_step_probe.py > $O/step1_d$d.txt 2>&1; tail -12 $O/step1_d$d.txt
done
for d in 0 1 0 1; do
VLNI_GEMM_DESYNC=$d $T 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/bench_d$d.json 2> $O/bench_d$d.err; cut -c1-330 $O/bench_d$d.json
done
