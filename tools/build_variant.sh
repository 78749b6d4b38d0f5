#!/bin/bash
# Builds an A/B copy of the library with extra hipcc flags for ONE source: tools/build_variant.sh <out.so> <source.hip> <flags...>
# (the other objects are the product build's). Load it with VLNI_LIB_PATH=<out.so>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; SRC=$2; shift 2
mkdir -p $R/vln-imagine_amd/build/variants
OBJ=$R/vln-imagine_amd/build/variants/$(basename $OUT .so)_$(basename $SRC .hip).o
EXTRA=""; if [ "$SRC" == "attention.hip" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi    # as vln-imagine_amd/build.py:EXTRA
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $EXTRA "$@" -c $R/vln-imagine_amd/csrc/$SRC -o $OBJ
OBJS=""
for s in api gemm layernorm elementwise attention graphmap blocks; do
  if [ "$s.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $R/vln-imagine_amd/build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo built $OUT
