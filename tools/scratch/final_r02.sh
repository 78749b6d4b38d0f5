#!/bin/bash
# round-2 artefacts: bench lines (HAMT, DUET), kernel-trace stats of both, PMC passes of the HAMT step
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
python3 bench.py --model duet > $O/bench_duet.json 2> $O/bench_duet.err
echo duet bench done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hamt -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-parity > $O/bench_prof_hamt.json 2> $O/bench_prof_hamt.err
python3 $R/tools/trace_breakdown.py $O/hamt 6 > $O/breakdown_hamt.txt || true
rm -f $O/hamt/*/*_kernel_trace.csv
echo hamt trace done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/duet -- python3 $R/bench.py --model duet --no-cpu-baseline --no-extras --no-parity > $O/bench_prof_duet.json 2> $O/bench_prof_duet.err
python3 $R/tools/trace_breakdown.py $O/duet 6 > $O/breakdown_duet.txt || true
rm -f $O/duet/*/*_kernel_trace.csv
echo duet trace done
B="python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.json 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
echo pmc done
cd $R && python3 tools/summarize_pmc.py $O/fetch $O/write $O/sq r02 > $O/pmc_summary.txt
cp profiles/r02_pmc_traffic.json profiles/r02_pmc_sq.md $O/
rm -rf $O/fetch $O/write $O/sq
tail -3 $O/breakdown_hamt.txt; cut -c1-300 $O/bench.json
