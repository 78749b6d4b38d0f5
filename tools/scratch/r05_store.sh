#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5v; mkdir -p $O; cd $R
python -m pytest tests/test_trainer_gpu.py -q -x -k "store_mode or loss_scaling" > $O/t_new.log 2>&1; tail -3 $O/t_new.log
python -m pytest tests/test_trainer_gpu.py tests/test_buckets_gpu.py tests/test_hamt_gpu.py tests/test_duet_gpu.py tests/test_tape_gpu.py tests/test_dp_gpu.py tests/test_dropout_gpu.py -q > $O/t_models.log 2>&1; tail -8 $O/t_models.log
A="--steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do for fz in 0 1; do
VLNI_BENCH_TRACEBACK=1 VLNI_STORE_PARTS=$fz python bench.py $A > $O/b_hamt_${fz}_$i.json 2> $O/b_hamt_${fz}_$i.err
VLNI_BENCH_TRACEBACK=1 VLNI_STORE_PARTS=$fz python bench.py --model duet $A > $O/b_duet_${fz}_$i.json 2> $O/b_duet_${fz}_$i.err
done; done
python - <<'PY'
import json
for n in ("hamt_0_1","hamt_1_1","hamt_0_2","hamt_1_2","duet_0_1","duet_1_1","duet_0_2","duet_1_2"):
    try:
        d=json.load(open(f"gpurun_out/r5v/b_{n}.json")); print(n, d["ms_per_step"], d.get("ms_per_step_median"), d["config"].get("launch","")[:40])
    except Exception as e: print(n, "failed", e)
PY
for m in hamt; do
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python3 bench.py --model $m --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof_$m.json 2> $O/prof_$m.err
python3 tools/step_profile.py $O/trace_$m $O/prof_$m.json r05x_$m 3 > $O/step_$m.md 2>/dev/null; grep -E "launches/step|reduce_parts|sumsq|FillFunctor|adamw" $O/step_$m.md
rm -rf $O/trace_$m
done
