#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
python tools/dropin_probe.py hamt --profile > $O/prof_hamt.log 2>&1; grep -A1 "drop-in eager" $O/prof_hamt.log
python tools/dropin_probe.py duet --profile > $O/prof_duet.log 2>&1; grep -A1 "drop-in eager" $O/prof_duet.log
