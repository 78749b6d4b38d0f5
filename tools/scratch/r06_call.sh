#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=$R/vln-imagine_amd/build/variants; cd $R
echo "== base"; GRAPH=1 timeout -k 10 200 python3 tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -8
for n in F5 F6; do echo "== $n"; VLNI_LIB_PATH=$V/lib_$n.so timeout -k 10 200 python3 tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -8; done
