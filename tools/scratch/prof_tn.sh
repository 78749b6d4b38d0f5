#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-roofline --steps 4 --warmup 2 > /tmp/pt.json 2> /tmp/pt.err
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Z"], r["Workgroup_Size_X"]) for r in csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if "adamw" in r[2]]
seg = rows[idx[-2] + 1: idx[-1] + 1]          # last graph-replayed step
tn = [(e - s, k, gx, gz, wx) for s, e, k, gx, gz, wx in seg if "gemm_tn" in k or "reduce_parts" in k]
print("launches", len(tn), "total ms", sum(t[0] for t in tn) / 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for d, k, gx, gz, wx in tn:
    name = "ring" if "ring" in k else "big" if "tn_big" in k else "glds" if "glds" in k else "reduce" if "reduce" in k else "tn"
    key = (name, int(gx) // int(wx), gz)
    agg[key][0] += 1; agg[key][1] += d
for key, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:7s} blocks {key[1]:5d} z {key[2]:>3s}: {n:3d} launches, {d / n / 1e3:7.1f} us avg, {d / 1e6:6.3f} ms")
PY
