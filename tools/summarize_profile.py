"""Turns a rocprofv3 --kernel-trace --stats output dir into profiles/<tag>_kernel_stats.{csv,md} (+ copies the bench line)."""
import csv, glob, json, shutil, sys, os
prof_dir, bench_json, prof_json, tag = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = glob.glob(os.path.join(prof_dir, "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
line = json.loads(open(prof_json).read())
steps = sum(int(r["Calls"]) for r in rows if "adamw" in r["Name"]) or (line["steps"] + line["warmup"] + 1)   # one AdamW launch per step
nt = [r for r in rows if any(k in r["Name"] for k in ("gemm_nt", "gemm_nn", "gemm_pk"))]
nt_calls = sum(int(r["Calls"]) for r in nt); nt_ns = sum(float(r["TotalDurationNs"]) for r in nt)
out = [f"# rocprofv3 --kernel-trace --stats ({tag})", "",
       "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-parity` "  + (sys.argv[5] + " " if len(sys.argv) > 5 else "") +
       "(1x MI355X, bf16, B=64, T=6, all-trainable, step-by-step calls, hipGraph replay; per-step figures = totals / number of AdamW "
       "launches; the first warm-up step also holds the per-shape GEMM autotune trials, ~8 variants x 3 launches per shape)", "",
       f"bench line under the profiler: {line['value']} episodes/s, {line['ms_per_step']} ms/step",
       f"kernel time total {tot/1e6:.1f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step",
       f"dominant kernel family gemm_pk* / gemm_nt* / gemm_nn* (one contraction: the persistent 128x128 kernel, thirteen other NT pipelines, the transposing-read dgrad forms): {nt_calls/steps:.0f} launches/step, average "
       f"{nt_ns/nt_calls/1e3:.1f} us per launch (rocprof, all launches incl. autotune trials) vs roofline.avg_launch_us "
       f"{line['roofline']['avg_launch_us']} us (HIP events around each launch of one eager step inside bench.py, behind a 50-ms spin "
       "kernel so that the pairs bracket the kernels only; each pair still adds ~2 us)", "",
       "| kernel | calls/step | ms/step | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:28]:
    out.append(f"| `{r['Name'][:96]}` | {int(r['Calls'])/steps:.0f} | {float(r['TotalDurationNs'])/1e6/steps:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
open(os.path.join(root, "profiles", f"{tag}_kernel_stats.md"), "w").write("\n".join(out) + "\n")
shutil.copy(f, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
if os.path.exists(bench_json):
    shutil.copy(bench_json, os.path.join(root, "profiles", f"{tag}_bench.json"))
print("\n".join(out[:24]))
