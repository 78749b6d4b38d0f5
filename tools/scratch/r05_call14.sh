#!/bin/bash
# round 5, call 14: ghost-pass clones avoided (HAMT), DUET masks once per episode: tests + both bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
python -m pytest tests/test_tape_gpu.py tests/test_buckets_gpu.py -q -x > $O/t_tape.log 2>&1; tail -2 $O/t_tape.log
python -m pytest tests/test_hamt_gpu.py tests/test_duet_gpu.py -q -x -k "reference_golden and (taped or graph)" > $O/t_gold.log 2>&1; tail -2 $O/t_gold.log
A="--steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
python bench.py $A > $O/b_hamt_$i.json 2> $O/b_hamt_$i.err
python bench.py --model duet $A > $O/b_duet_$i.json 2> $O/b_duet_$i.err
done
python - <<'PY'
import json
for n in ("hamt_1","hamt_2","duet_1","duet_2"):
    try:
        d=json.load(open(f"gpurun_out/r5q/b_{n}.json")); print(n, d["ms_per_step"], d.get("ms_per_step_median"))
    except Exception as e: print(n, "failed", e)
PY
