#!/bin/bash
# round 5, call 6: issue order of the step's two concurrent calls (history first / visual first) + trace of the better one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
for i in 1 2; do
VLNI_HISTORY_AFTER=0 python bench.py $A > $O/b_first_$i.json 2> $O/b_first_$i.err; echo first done
VLNI_HISTORY_AFTER=1 python bench.py $A > $O/b_after_$i.json 2> $O/b_after_$i.err; echo after done
VLNI_HISTORY_AFTER=1 VLNI_LOCKSTEP_HISTORY=1 python bench.py $A > $O/b_lock_$i.json 2> $O/b_lock_$i.err; echo lock done
done
python - <<'PY'
import json
for n in ("first_1","after_1","lock_1","first_2","after_2","lock_2"):
    try:
        d=json.load(open(f"gpurun_out/r5g/b_{n}.json")); print(n, d["ms_per_step"])
    except Exception as e: print(n, "failed", e)
PY
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof.json 2> $O/prof.err
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(os.path.join(R, "gpurun_out/r5g/trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-1400:]
t0 = int(keep[0]["Start_Timestamp"])
with open(os.path.join(R, "gpurun_out/r5g/last_step.tsv"), "w") as fh:
    for r in keep:
        fh.write("\t".join([str((int(r["Start_Timestamp"]) - t0) / 1000.0), str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")]) + "\n")
PY
rm -rf $O/trace
