#!/bin/bash
# round 5, call 11: same-box traces of the step with per-step history calls beside the steps (rounds 2-4) and with the batched call up front
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
python -m pytest tests/test_tape_gpu.py -q -x -s -k "bf16_full_width" > $O/t_tape.log 2>&1; grep -E "bf16 vs fp32|passed|failed" $O/t_tape.log
python -m pytest tests/test_fulldepth_gpu.py -q -x -s -k "timed_path and hamt" > $O/t_full.log 2>&1; grep -E "^\[hamt|passed|failed" $O/t_full.log
A="--steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2 3; do
VLNI_HISTORY_UPFRONT=0 VLNI_HISTORY_AFTER=0 python bench.py $A > $O/b_old_$i.json 2> $O/b_old_$i.err
VLNI_HISTORY_UPFRONT=1 python bench.py $A > $O/b_up_$i.json 2> $O/b_up_$i.err
done
python - <<'PY'
import json
for n in ("old_1","up_1","old_2","up_2","old_3","up_3"):
    try:
        d=json.load(open(f"gpurun_out/r5l/b_{n}.json")); print(n, d["ms_per_step"], d.get("ms_per_step_median"))
    except Exception as e: print(n, "failed", e)
PY
for m in old up; do
if [ $m = old ]; then export VLNI_HISTORY_UPFRONT=0 VLNI_HISTORY_AFTER=0; else export VLNI_HISTORY_UPFRONT=1; unset VLNI_HISTORY_AFTER; fi
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof_$m.json 2> $O/prof_$m.err
python3 - $m <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; m=sys.argv[1]
f = glob.glob(os.path.join(R, f"gpurun_out/r5l/trace_{m}/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-1400:]
t0 = int(keep[0]["Start_Timestamp"])
with open(os.path.join(R, f"gpurun_out/r5l/last_step_{m}.tsv"), "w") as fh:
    for r in keep:
        fh.write("\t".join([str((int(r["Start_Timestamp"]) - t0) / 1000.0), str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")]) + "\n")
PY
rm -rf $O/trace_$m
python3 tools/trace_segments.py $O/last_step_$m.tsv
done
