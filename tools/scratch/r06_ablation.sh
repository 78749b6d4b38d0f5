#!/bin/bash
# round 6: bf16 / fp16 ablation lines + the three 300-step training runs (tools/bf16_ablation.py); then the remaining GPU tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6abl
mkdir -p $O; cd $R
T="timeout -k 10"
run() { tag=$1; shift; env "$@" $T 400 python3 tools/bf16_ablation.py ablate "$tag" 2> $O/$tag.err | grep "^{" > $O/$tag.json; cut -c1-300 $O/$tag.json; }
run a_default VLNI_ABL_DTYPE=bf16
run b_gelu_recompute VLNI_ABL_DTYPE=bf16 VLNI_GELU_STORE_GRAD=0
run c_lang_qkv_per_step VLNI_ABL_DTYPE=bf16 VLNI_LANG_QKV_ONCE=0
run d_stepwise VLNI_ABL_DTYPE=bf16 VLNI_ABL_MODE=stepwise
run e_batch8 VLNI_ABL_DTYPE=bf16 VLNI_ABL_BATCH=8
run f_fp16 VLNI_ABL_DTYPE=fp16
for d in fp32 bf16 fp16; do $T 900 python3 tools/bf16_ablation.py train $d 2> $O/train_$d.err | grep "^{" > $O/t_$d.json; cut -c1-200 $O/t_$d.json; done
python3 tools/bf16_ablation.py report $O > $O/r06_bf16_ablation.md; head -30 $O/r06_bf16_ablation.md | cut -c1-250
$T 600 python3 -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "stale_weight" 2>&1 | tail -3
