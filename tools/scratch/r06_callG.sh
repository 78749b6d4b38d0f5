#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for f in hamt duet; do FAMILY=$f timeout -k 10 300 python3 tools/scratch/native_grads.py 2>&1 | grep -v "amdgpu.ids" | tail -12; done
