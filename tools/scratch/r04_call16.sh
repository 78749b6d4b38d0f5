#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c16
mkdir -p $O
cd $R
A="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --kernel-trace --output-format csv -d $O/trace_hamt -o t -- python3 bench.py $A > $O/prof_hamt.json 2> $O/prof_hamt.err
python3 tools/step_profile.py $O/trace_hamt $O/prof_hamt.json r04tmp 6 > $O/breakdown_hamt.txt
rm -rf $O/trace_hamt profiles/r04tmp*
sed -n 1,60p $O/breakdown_hamt.txt | cut -c1-160
