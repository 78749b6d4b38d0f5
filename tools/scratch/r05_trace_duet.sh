#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --model duet --steps 4 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $O/prof.json 2> $O/prof.err
python3 tools/step_profile.py $O/trace $O/prof.json r05_duet_x 3 > $O/breakdown.txt 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(os.path.join(R, "gpurun_out/r5p/trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-1900:]
t0 = int(keep[0]["Start_Timestamp"])
with open(os.path.join(R, "gpurun_out/r5p/last_step.tsv"), "w") as fh:
    for r in keep:
        fh.write("\t".join([str((int(r["Start_Timestamp"]) - t0) / 1000.0), str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")]) + "\n")
PY
rm -rf $O/trace
