"""profiles/<tag>_step_breakdown.md + profiles/<tag>_kernel_stats.csv from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
ONLY the last N graph-replayed steps of the timed region (delimited by the AdamW launches), so no autotune trial, capture warm-up,
instrumented roofline step or spin kernel is counted. Per kernel family and per kernel: launches / step,
ms / step, average us; the span of the N steps against the sum of kernel times (> 1 where the side stream overlaps).
usage: python tools/step_profile.py <rocprof dir> <bench json line> <tag> [N=6]"""
import collections
import csv
import glob
import json
import os
import re
import sys

d, bench_json, tag = sys.argv[1:4]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 6
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
line = json.loads([l for l in open(bench_json).read().splitlines() if l.startswith("{")][0])
# the timed region = the `steps` AdamW launches before the instrumented roofline pass (bench.py runs the roofline, parity and extras AFTER it)
marks = [i for i, r in enumerate(rows) if "adamw" in r[2]]
warm, steps = line["warmup"], line["steps"]
timed_last = warm + 1 + 3 + steps - 1          # warm-up steps, 1 capture warm-up + 3 replays (bench.measure), then `steps` timed replays
timed_last = min(timed_last, len(marks) - 1)
a, b = marks[timed_last - N], marks[timed_last]
seg = rows[a + 1:b + 1]
FAM = (("gemm_p8h", "GEMM 256x128 loader-wave kernel (fwd / dgrad)"), ("gemm_p8", "GEMM 256x256 8-phase kernel (fwd / dgrad)"),
       ("gemm_pk", "GEMM persistent 128x128 (fwd / dgrad)"), ("gemm_nt", "GEMM other NT pipelines (fwd / dgrad)"), ("gemm_nn", "GEMM [K,N] dgrad"),
       ("gemm_tn", "weight-gradient GEMMs"), ("reduce_parts", "weight-gradient partial reduction"), ("attn_bwd", "attention backward"),
       ("attn_fwd", "attention forward"), ("ln_bwd", "LayerNorm backward"), ("ln_fwd", "LayerNorm forward (+ sum_ln)"), ("adamw", "clip + AdamW"),
       ("sumsq", "gradient sum of squares"), ("transpose", "W^T shadows (batched transpose)"), ("at::native", "torch-native glue"),
       ("Functor", "torch-native glue"), ("CatArray", "torch-native glue"))


def fam(k):
    for key, name in FAM:
        if key in k:
            return name
    return "other vlni kernels (embeddings, heads, losses, dropout, casts)"


agg, per = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    agg[fam(k)][0] += 1; agg[fam(k)][1] += e - s
    m_ = re.search(r"(\w+_kernel)(<[^(]*>)?", k)
    short = (m_.group(1) + (m_.group(2) or "")[:40]) if m_ else re.sub(r"\(.*", "", k).replace("void ", "")[-70:]
    per[short][0] += 1; per[short][1] += e - s
tot = sum(v[1] for v in agg.values())
span = seg[-1][1] - seg[0][0]
out = [f"# Where a step goes ({tag}): the last {N} graph-replayed steps of the timed region", "",
       f"source: `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py {line.get('argv', '')}` on one MI355X; bench line under the profiler: "
       f"{line['value']} {line['unit']}, {line['ms_per_step']} ms/step ({line['config']['workload']}).", "",
       f"span {span / 1e6 / N:.2f} ms/step, sum of kernel times {tot / 1e6 / N:.2f} ms/step (the difference is what runs concurrently on the side stream), "
       f"{len(seg) / N:.0f} launches/step.", "", "| kernel family | launches / step | ms / step | average us | share of kernel time |", "|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append(f"| {k} | {v[0] / N:.1f} | {v[1] / 1e6 / N:.3f} | {v[1] / v[0] / 1e3:.1f} | {100 * v[1] / tot:.1f} % |")
out += ["", "| kernel | launches / step | ms / step | average us |", "|---|---|---|---|"]
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
    out.append(f"| `{k}` | {v[0] / N:.1f} | {v[1] / 1e6 / N:.3f} | {v[1] / v[0] / 1e3:.1f} |")
open(os.path.join(root, "profiles", f"{tag}_step_breakdown.md"), "w").write("\n".join(out) + "\n")
with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w") as g:
    g.write("kernel,launches_per_step,ms_per_step,avg_us\n")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        g.write(f"\"{k}\",{v[0] / N:.2f},{v[1] / 1e6 / N:.4f},{v[1] / v[0] / 1e3:.2f}\n")
print("\n".join(out[:30]))
