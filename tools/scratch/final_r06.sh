#!/bin/bash
# round-6 evidence: default bench lines (HAMT with the CPU baseline + DUET configs[3] inside it, DUET full line), kernel traces of both, PMC passes
# that LOAD the timed run's kernel choices (--dump-tune / --load-tune), the per-shape GEMM table beside the vendor library (tools/gemm_shapes.py),
# the 1-rank RCCL rehearsal of the multi-GPU line's self-checks. Each rocprofv3 run is its own pass; counters never share a run with a trace domain.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final
mkdir -p $O
cd $R
VLNI_GEMM_SHAPES_OUT=$O/shapes_hamt.json python3 bench.py --dump-tune $O/tune_hamt.json > $O/bench_hamt.json 2> $O/bench_hamt.err
echo "hamt bench done"
VLNI_GEMM_SHAPES_OUT=$O/shapes_duet.json python3 bench.py --model duet --dump-tune $O/tune_duet.json > $O/bench_duet.json 2> $O/bench_duet.err
echo "duet bench done"
A="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --kernel-trace --output-format csv -d $O/trace_hamt -o t -- python3 bench.py $A --load-tune $O/tune_hamt.json > $O/prof_hamt.json 2> $O/prof_hamt.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace_duet -o t -- python3 bench.py --model duet $A --load-tune $O/tune_duet.json > $O/prof_duet.json 2> $O/prof_duet.err
python3 tools/step_profile.py $O/trace_hamt $O/prof_hamt.json r06 6 > $O/breakdown_hamt.txt
python3 tools/step_profile.py $O/trace_duet $O/prof_duet.json r06_duet 6 > $O/breakdown_duet.txt
echo "traces done"
P="--steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py $P --load-tune $O/tune_hamt.json > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py $P --load-tune $O/tune_hamt.json > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/sq -- python3 bench.py $P --load-tune $O/tune_hamt.json > /dev/null 2> $O/sq.err
python3 tools/summarize_pmc.py $O/fetch $O/write $O/sq r06 > $O/pmc_hamt.txt
echo "hamt pmc done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_d -- python3 bench.py --model duet $P --load-tune $O/tune_duet.json > /dev/null 2> $O/fetch_d.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_d -- python3 bench.py --model duet $P --load-tune $O/tune_duet.json > /dev/null 2> $O/write_d.err
python3 tools/summarize_pmc.py $O/fetch_d $O/write_d - r06_duet > $O/pmc_duet.txt
echo "duet pmc done"
python3 tools/pmc_vs_trace.py r06 > $O/launch_check.txt || true
cat $O/launch_check.txt
for f in hamt duet; do
rocprofv3 --kernel-trace --output-format csv -d $O/trace_shapes_$f -o t -- python3 tools/gemm_shapes.py run $O/shapes_$f.json $O/shapes_run_$f.json > $O/shapes_run_$f.txt 2>&1
done
python3 tools/gemm_shapes.py report $O/shapes_run_hamt.json $O/trace_shapes_hamt "Forward / dgrad GEMM launches of one HAMT step (BASELINE.json configs[1]) by shape, beside the vendor library (r06)" > $O/r06_gemm_shapes.md
python3 tools/gemm_shapes.py report $O/shapes_run_duet.json $O/trace_shapes_duet "Forward / dgrad GEMM launches of one DUET step (BASELINE.json configs[3]) by shape, beside the vendor library (r06)" > $O/r06_duet_gemm_shapes.md
python3 tools/gemm_traffic.py $O/shapes_run_hamt.json profiles/r06_pmc_traffic.json "Fabric traffic of the forward / dgrad GEMM family, HAMT step (r06)" > $O/r06_gemm_traffic.md
python3 tools/gemm_traffic.py $O/shapes_run_duet.json profiles/r06_duet_pmc_traffic.json "Fabric traffic of the forward / dgrad GEMM family, DUET step (r06)" > $O/r06_duet_gemm_traffic.md
echo "shapes done"
VLNI_FORCE_COLLECTIVES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/bench_forced_rccl.json 2> $O/bench_forced_rccl.err
cp profiles/r06* $O/ 2>/dev/null || true
cp $O/bench_hamt.json profiles/r06_bench.json; cp $O/bench_duet.json profiles/r06_bench_duet.json
cp profiles/r06_bench*.json $O/
rm -rf $O/trace_hamt $O/trace_duet $O/fetch $O/write $O/sq $O/fetch_d $O/write_d $O/trace_shapes_hamt $O/trace_shapes_duet
head -30 $O/breakdown_hamt.txt
