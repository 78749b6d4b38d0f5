"""Reads a rocprofv3 kernel_trace.csv: busy time vs idle gaps between consecutive kernels (gaps < 1 ms only, i.e. inside steps)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 40000          # last N kernels ~ the timed graph replays
rows = rows[-tail:]
busy = sum(e - s for s, e, _ in rows)
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 < g[0] < 1_000_000]
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)} span {span/1e6:.1f} ms busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), gaps<1ms total {sum(g[0] for g in small)/1e6:.1f} ms, mean gap {sum(g[0] for g in small)/max(1,len(small))/1e3:.2f} us")
by = collections.Counter()
for g, a, b in small:
    by[a[:60]] += g
for k, v in by.most_common(12):
    print(f"  gap after {k:60s} {v/1e6:7.2f} ms")
