"""Do the counter passes describe the timed program? Compares the per-family GEMM launch counts of the rocprofv3 --pmc passes
(profiles/<tag>_pmc_traffic.json, eager --no-graph --load-tune step) with those of the kernel trace of the graph-replayed, timed steps
(profiles/<tag>_kernel_stats.csv). usage: python tools/pmc_vs_trace.py <tag>"""
import csv, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")))["per_family_kb_per_launch"]
trace = {}
for r in csv.DictReader(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))):
    for fam in ("gemm_p8h", "gemm_p8", "gemm_pk", "gemm_nt_big", "gemm_nt_glds", "gemm_nt_kernel", "gemm_nn_glds", "gemm_tn_ring", "gemm_tn_big", "gemm_tn_glds", "gemm_tn_bf16",
                "reduce_parts", "attn_bwd", "attn_fwd", "ln_bwd", "ln_fwd", "adamw"):
        if fam in r["kernel"]:
            trace[fam] = trace.get(fam, 0.0) + float(r["launches_per_step"])
            break
pm = {}
for k, v in pmc.items():
    k2 = "attn_bwd" if k.startswith("attn_bwd") else "attn_fwd" if k.startswith("attn_fwd") else k
    pm[k2] = pm.get(k2, 0) + v["launches"]
bad = 0
print(f"| kernel family | launches / step, timed trace | launches, counter pass |\n|---|---|---|")
for k in sorted(set(trace) | set(pm)):
    a, b = trace.get(k, 0.0), pm.get(k, 0)
    flag = "" if abs(a - b) < 0.5 else "  <-- differs"
    bad += bool(flag)
    print(f"| `{k}` | {a:.0f} | {b} |{flag}")
print("counter passes run the timed step's kernel mix" if not bad else f"{bad} families differ")
sys.exit(1 if bad else 0)
