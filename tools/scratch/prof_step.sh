#!/bin/bash
# kernel trace of the default bench step -> gpurun_out/<out>/{breakdown.txt, stats dir}
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/$out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-parity --no-cpu-baseline --steps 10 --warmup 3 "$@" > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/bench_prof.err
python3 $GRAFT_REPO_ROOT/tools/trace_breakdown.py /tmp/prof_step 8 > $GRAFT_REPO_ROOT/$out/breakdown.txt 2>&1
mkdir -p $GRAFT_REPO_ROOT/$out/prof && cp /tmp/prof_step/*/*kernel_stats.csv $GRAFT_REPO_ROOT/$out/prof/ 2>/dev/null
cat $GRAFT_REPO_ROOT/$out/breakdown.txt
