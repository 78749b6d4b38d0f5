#!/bin/bash
cd $GRAFT_REPO_ROOT
VLNI_LIB_PATH=$GRAFT_REPO_ROOT/vln-imagine_amd/build/variants/st.so timeout -k 10 300 python3 tools/attn_stamps.py 0.1 2>&1 | grep -v "amdgpu\|alive" | head -12
