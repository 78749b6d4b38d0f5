#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in st2 st3; do
  echo "== variant $v"
  VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/$v.so timeout -k 10 300 python3 tools/attn_stamps.py 0.0 2>&1 | grep -v amdgpu
done
VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/st2.so timeout -k 10 300 python3 tools/attn_stamps.py 0.1 2>&1 | grep -v amdgpu | head -14
