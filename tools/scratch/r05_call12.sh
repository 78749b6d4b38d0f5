#!/bin/bash
# round 5, call 12: DUET - the T panorama calls of a teacher-forced episode as one batched call up front (tests, bench A/B)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
python -m pytest tests/test_tape_gpu.py tests/test_buckets_gpu.py -q -x -k "duet" > $O/t_tape.log 2>&1; tail -3 $O/t_tape.log
python -m pytest tests/test_duet_gpu.py -q -x -k "reference_golden and (taped or graph)" > $O/t_duet.log 2>&1; tail -3 $O/t_duet.log
python -m pytest tests/test_fulldepth_gpu.py -q -x -s -k "timed_path" > $O/t_full.log 2>&1; grep -E "^\[|passed|failed" $O/t_full.log
A="--model duet --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
VLNI_PANORAMA_UPFRONT=0 python bench.py $A > $O/b_old_$i.json 2> $O/b_old_$i.err
VLNI_PANORAMA_UPFRONT=1 python bench.py $A > $O/b_up_$i.json 2> $O/b_up_$i.err
done
python - <<'PY'
import json
for n in ("old_1","up_1","old_2","up_2"):
    try:
        d=json.load(open(f"gpurun_out/r5m/b_{n}.json")); print(n, d["ms_per_step"], d.get("ms_per_step_median"))
    except Exception as e: print(n, "failed", e)
PY
