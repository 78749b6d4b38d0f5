mkdir -p gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr8 -- python3 $GRAFT_REPO_ROOT/tools/attn_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r4g/ap8.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr8 -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/r4g/ap8_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "attn" in n:
        seq.append((n[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
# consecutive runs of the same kernel = one probe line
prev, run = None, []
for s in seq + [(None,)]:
    if s[0] != prev and run:
        d = sorted(x[1] for x in run)
        print(f"{prev:60s} n={len(run):3d} median {d[len(d)//2]:7.1f} us min {d[0]:7.1f}  grid {run[0][2]} wg {run[0][3]} lds {run[0][4]} vgpr {run[0][5]}")
        run = []
    prev = s[0]
    if s[0] is not None:
        run.append(s)
PY
