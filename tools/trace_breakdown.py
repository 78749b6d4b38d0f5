"""Per-kernel-family time per step from a rocprofv3 kernel_trace.csv of bench.py (steps delimited by the AdamW launches)."""
import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv") + glob.glob(sys.argv[1] + "/*_kernel_trace.csv"))[0]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "adamw" in r[2]]
a, b = idx[-nsteps - 1], idx[-1]
seg = rows[a + 1:b + 1]
FAM = ("gemm_p8h", "gemm_p8", "gemm_pk", "gemm_nt", "gemm_nn", "gemm_tn", "reduce_parts", "attn_bwd", "attn_fwd", "ln_bwd", "ln_fwd", "adamw", "transpose", "cast_kernel", "scatter_add",
       "smallk", "FillFunctor", "CatArray", "CUDAFunctor_add", "direct_copy", "MulFunctor", "reduce_kernel", "copyBuffer")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    key = next((x for x in FAM if x in k), k[:48])
    agg[key][0] += 1
    agg[key][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"{nsteps} steps: kernel time {tot/1e6/nsteps:.2f} ms/step, span {(seg[-1][1]-seg[0][0])/1e6/nsteps:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {k:50s} {v[0]/nsteps:7.1f} calls {v[1]/1e6/nsteps:7.3f} ms  avg {v[1]/v[0]/1e3:7.1f} us")
