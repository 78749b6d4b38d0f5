#!/bin/bash
# kernel trace of the default bench step (csv kept under gpurun_out/r5f): the launch sequence of one replayed step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof.json 2> $O/prof.err
ls -la $O/trace/*/ | head; python3 tools/step_profile.py $O/trace $O/prof.json r05x 3 > $O/breakdown.txt 2>&1; head -50 $O/breakdown.txt
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5f/trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~1300 launches (one step and a bit), compact columns
keep = rows[-1400:]
t0 = int(keep[0]["Start_Timestamp"])
with open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5f/last_step.tsv"), "w") as fh:
    for r in keep:
        fh.write("\t".join([str((int(r["Start_Timestamp"]) - t0) / 1000.0), str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")]) + "\n")
PY
rm -rf $O/trace
