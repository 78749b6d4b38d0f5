#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
for o in serial side_first main_first interleaved; do python3 tools/graph_fork_probe.py $o 12 120; done
for o in side_first main_first interleaved; do
rocprofv3 --kernel-trace --output-format csv -d $O/tr_$o -o t -- python3 tools/graph_fork_probe.py $o 12 120 > /dev/null 2>&1
python3 - $o <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; o=sys.argv[1]
f = glob.glob(os.path.join(R, f"gpurun_out/r5j/tr_{o}/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-135:]; t0 = int(keep[0]["Start_Timestamp"])
print("==", o)
last=None
for r in keep:
    q=r.get("Queue_Id"); nm="LONG" if "Cijk" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"].lower() else "s"
    if nm=="LONG" or last!=("s",q):
        print(f"{(int(r['Start_Timestamp'])-t0)/1000:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:7.1f} q{q} {nm}")
    last=(nm,q)
PY
rm -rf $O/tr_$o
done
