"""Durations of every launch of kernels matching a name fragment within the last bench step of a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?")) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "adamw" in r[2]]
seg = rows[idx[-2] + 1:idx[-1] + 1]
for frag in sys.argv[2:]:
    sel = [(e - s) / 1e3 for s, e, k, g in seg if frag in k]
    print(frag, len(sel), "calls:", " ".join(f"{x:.0f}" for x in sel), "us; total", f"{sum(sel):.0f}")
