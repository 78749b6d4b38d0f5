#!/bin/bash
# round 5, call 8: all history calls of a teacher-forced episode up front on the second stream (tests, bench A/B, trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
python -m pytest tests/test_tape_gpu.py tests/test_buckets_gpu.py tests/test_dropout_gpu.py -q -x > $O/t_tape.log 2>&1; tail -3 $O/t_tape.log
python -m pytest tests/test_hamt_gpu.py -q -x -k "reference_golden and (taped or graph)" > $O/t_hamt.log 2>&1; tail -3 $O/t_hamt.log
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
VLNI_HISTORY_UPFRONT=1 VLNI_OVERLAP_HISTORY=0 python bench.py $A > $O/b_single_$i.json 2> $O/b_single_$i.err; echo single done
VLNI_HISTORY_UPFRONT=0 VLNI_HISTORY_AFTER=0 python bench.py $A > $O/b_old_$i.json 2> $O/b_old_$i.err; echo old done
VLNI_HISTORY_UPFRONT=1 python bench.py $A > $O/b_up_$i.json 2> $O/b_up_$i.err; echo upfront done
done
python - <<'PY'
import json
for n in ("old_1","up_1","single_1","old_2","up_2","single_2"):
    try:
        d=json.load(open(f"gpurun_out/r5k/b_{n}.json")); print(n, d["ms_per_step"])
    except Exception as e: print(n, "failed", e)
PY
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof.json 2> $O/prof.err
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(os.path.join(R, "gpurun_out/r5k/trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-1400:]
t0 = int(keep[0]["Start_Timestamp"])
with open(os.path.join(R, "gpurun_out/r5k/last_step.tsv"), "w") as fh:
    for r in keep:
        fh.write("\t".join([str((int(r["Start_Timestamp"]) - t0) / 1000.0), str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")]) + "\n")
PY
rm -rf $O/trace
