"""Segments of one replayed HAMT step from a `last_step.tsv` (tools/scratch/r05_trace.sh): episode start -> first step, the T steps, loss + backward, the
weight-gradient flush, the optimizer; kernel-time sums, launch counts, and per-queue idle time inside each segment.
usage: python tools/trace_segments.py gpurun_out/<dir>/last_step.tsv"""
import sys

rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
idx = [i for i, r in enumerate(rows) if "adamw" in r[4]]
s = rows[idx[-2] + 1:idx[-1] + 1]
t0 = float(s[0][0])
ev = [(float(r[0]) - t0, float(r[1]), r[2], r[4]) for r in s]
rd = [x[0] + x[1] for x in ev if "rowdot_fwd" in x[3]]
first_vis = None
for x in ev:                      # the first step starts where the first dual attention / p8h<1> of the cross-modal layers appears: use the cast before it
    if "attn_fwd_bf16_dual" in x[3]:
        first_vis = x[0]
        break
ring = [x[0] for x in ev if "gemm_tn" in x[3]]
red = [x[0] for x in ev if "reduce_parts" in x[3]]
marks = [("episode start (text encoder, history, aux head)", 0.0, first_vis)]
prev = first_vis
for i, e in enumerate(rd):
    marks.append((f"step {i}", prev, e))
    prev = e
marks.append(("loss + backward", prev, ring[0] if ring else prev))
marks.append(("weight-gradient flush + reduction", ring[0], red[-1] + 600 if red else ring[-1]))
marks.append(("rest (norm, AdamW)", marks[-1][2], ev[-1][0] + ev[-1][1]))
print(f"{len(ev)} launches, span {ev[-1][0] + ev[-1][1]:.0f} us")
for name, a, b in marks:
    seg = [x for x in ev if a <= x[0] < b]
    print(f"{name:50s} {b - a:8.0f} us  {len(seg):4d} launches  kernel-time sum {sum(x[1] for x in seg):8.0f} us  queues {sorted(set(x[2] for x in seg))}")
