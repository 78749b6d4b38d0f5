#!/bin/bash
# kernel trace of the default bench step (last 6 graph-replayed steps) -> profiles/r04_step_breakdown.md (+ glue kernel list)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4trace
mkdir -p $O
cd $R
A="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-parity --no-roofline"
rocprofv3 --kernel-trace --output-format csv -d $O/trace_hamt -o t -- python3 bench.py $A > $O/prof_hamt.json 2> $O/prof_hamt.err
python3 tools/step_profile.py $O/trace_hamt $O/prof_hamt.json r04tmp 6 > $O/breakdown_hamt.txt
cp profiles/r04tmp_step_breakdown.md $O/
python3 - <<'PY'
import csv, glob, collections, json, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r4trace"
f = (glob.glob(O + "/trace_hamt/*/*_kernel_trace.csv") + glob.glob(O + "/trace_hamt/*_kernel_trace.csv"))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if "adamw" in r[2]]
a, b = marks[-7], marks[-1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows[a + 1:b + 1]:
    if "vlni" in k or "_GLOBAL__N_" in k or "k_bf16" in k or "k_f16" in k:
        key = "VLNI " + k.split("(")[0][-60:]
    else:
        key = "ATEN " + k[:150]
    agg[key][0] += 1; agg[key][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] / 6e3 < 2: continue
    print(f"{v[0] / 6:6.1f} launches {v[1] / 6e3:8.1f} us/step  {k}")
PY
rm -rf $O/trace_hamt
