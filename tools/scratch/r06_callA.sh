#!/bin/bash
# round 6, call A: request-placement variants of the ring weight-gradient kernel, epilogue prefetch depths of the 256-wide NT kernels,
# in-kernel stamps of both, the step's GEMM shapes beside the vendor library (tools/gemm_shapes.py)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; V=$R/vln-imagine_amd/build/variants
mkdir -p $O; cd $R
T="timeout -k 10"
$T 300 python3 tools/ring_probe.py base > $O/ring_base.txt 2>&1; tail -2 $O/ring_base.txt
for n in A B C D; do VLNI_LIB_PATH=$V/lib_$n.so $T 200 python3 tools/ring_probe.py $n > $O/ring_$n.txt 2>&1; tail -2 $O/ring_$n.txt; done
for n in S0 S1; do STAMPS=1 VLNI_LIB_PATH=$V/lib_$n.so $T 200 python3 tools/ring_probe.py $n > $O/ring_$n.txt 2>&1; tail -3 $O/ring_$n.txt; done
for n in S0 S1; do VLNI_LIB_PATH=$V/lib_$n.so $T 200 python3 tools/p8_stamps.py $n > $O/p8_$n.txt 2>&1; cat $O/p8_$n.txt; done
T=6 GRAPH=1 STEP_KINDS=1 NT_VARIANTS=15,32 NN_VARIANTS=6 $T 300 python3 tools/gemm_step_probe.py > $O/step_base.txt 2>&1; tail -12 $O/step_base.txt
for n in A B C; do VLNI_LIB_PATH=$V/lib_$n.so T=6 GRAPH=1 STEP_KINDS=1 NT_VARIANTS=15,32 NN_VARIANTS=6 $T 300 python3 tools/gemm_step_probe.py > $O/step_$n.txt 2>&1; tail -12 $O/step_$n.txt; done
VLNI_GEMM_SHAPES_OUT=$O/shapes.json $T 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --dump-tune $O/tune.json > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-600
$T 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_shapes -o t -- python3 tools/gemm_shapes.py run $O/shapes.json $O/shapes_run.json > $O/shapes_run.txt 2>&1; tail -5 $O/shapes_run.txt
python3 tools/gemm_shapes.py report $O/shapes_run.json $O/trace_shapes "Forward / dgrad GEMM shapes of one HAMT step (r06, start of round)" > $O/r06_gemm_shapes_start.md
find $O/trace_shapes -name "*.csv" -size +20M -delete
$T 300 python3 -m pytest tests/test_edges_gpu.py tests/test_ops_gpu.py -q -x -k "embed or scatter or block or edge" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
