#!/bin/bash
# round-2 profile of the default bench line: kernel trace + stats (HAMT and DUET)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2p
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/hamt -- python3 bench.py --no-cpu-baseline --no-extras --no-parity > gpurun_out/r2p/bench_prof_hamt.json 2> gpurun_out/r2p/bench_prof_hamt.err
echo hamt done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/duet -- python3 bench.py --model duet --no-cpu-baseline --no-extras --no-parity > gpurun_out/r2p/bench_prof_duet.json 2> gpurun_out/r2p/bench_prof_duet.err
echo duet done
python tools/trace_breakdown.py gpurun_out/r2p/hamt 6 > gpurun_out/r2p/breakdown_hamt.txt
python tools/trace_breakdown.py gpurun_out/r2p/duet 6 > gpurun_out/r2p/breakdown_duet.txt
# keep the merged output small: the traces themselves stay on the box
rm -f gpurun_out/r2p/*/*/*_kernel_trace.csv
cat gpurun_out/r2p/breakdown_hamt.txt
