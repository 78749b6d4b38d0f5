#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f_final; mkdir -p $O; cd $R
python3 bench.py > $O/bench_hamt.json 2> $O/bench_hamt.err
python3 bench.py --model duet > $O/bench_duet.json 2> $O/bench_duet.err
tail -c 600 $O/bench_duet.err
