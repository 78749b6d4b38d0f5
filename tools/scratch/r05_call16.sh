#!/bin/bash
# round 5, call 16: eager (no captured graph) steps with and without the block-level calls: stepwise / taped, HAMT / DUET
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
python -m pytest tests/test_tape_gpu.py tests/test_hamt_gpu.py -q -x -k "taped_episode_equals or (reference_golden and c1_language)" > $O/t.log 2>&1; tail -2 $O/t.log
A="--steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline"
for b in 0 1; do for m in stepwise taped; do for f in hamt duet; do
VLNI_BLOCK_CALLS=$b python bench.py --model $f --mode $m $A > $O/b_${f}_${m}_$b.json 2> $O/b_${f}_${m}_$b.err
python - <<PY
import json
try: print("$f $m block_calls=$b", json.load(open("$O/b_${f}_${m}_$b.json"))["ms_per_step"])
except Exception as e: print("$f $m $b failed", e)
PY
done; done; done
