#!/bin/bash
# round-2 PMC passes of the bench step (each counter set its own run, no trace domains)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2q
mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.json 2> $O/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.json 2> $O/write.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
echo sq done
cd $R && python3 tools/summarize_pmc.py $O/fetch $O/write $O/sq r02 > $O/summary.txt
cp profiles/r02_pmc_traffic.json profiles/r02_pmc_sq.md $O/
rm -rf $O/fetch $O/write $O/sq
cat $O/summary.txt
