#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
python tools/dropin_probe.py hamt --profile > $O/hamt.log 2>&1; head -3 $O/hamt.log
python tools/dropin_probe.py duet --profile > $O/duet.log 2>&1; head -3 $O/duet.log
