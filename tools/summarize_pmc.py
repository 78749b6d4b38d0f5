"""Turns three rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ/GRBM counters, each its own run) into
profiles/<tag>_pmc_traffic.json (HBM bytes per launch of the NT GEMM family, gfx950 FETCH_SIZE correction applied) and
profiles/<tag>_pmc_sq.md (MFMA utilisation, LDS bank conflicts per kernel family).
usage: summarize_pmc.py <fetch_dir> <write_dir> <sq_dir> <tag>"""
import collections, csv, glob, json, os, sys
fetch_dir, write_dir, sq_dir, tag = sys.argv[1:5]          # sq_dir "-": traffic only
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d):
    """Counter rows of the LAST full step only (between the last two AdamW dispatches): the warm-up steps hold the per-shape GEMM
    autotune trials, which would weight the per-launch averages towards the large shapes."""
    f = (glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    marks = sorted({int(r["Dispatch_Id"]) for r in rows if "adamw" in r["Kernel_Name"]})
    lo, hi = marks[-2], marks[-1]
    return [r for r in rows if lo < int(r["Dispatch_Id"]) <= hi]


def fam(name):
    for k in ("gemm_p8h", "gemm_p8", "gemm_pk", "gemm_nt_big", "gemm_nt_glds", "gemm_nt_kernel", "gemm_nn_glds", "gemm_tn_ring", "gemm_tn_big", "gemm_tn_glds", "gemm_tn_bf16", "reduce_parts", "attn_bwd",
              "attn_fwd", "ln_bwd", "ln_fwd", "adamw"):
        if k in name:
            return k
    return None


def per_family(rows, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = fam(r["Kernel_Name"])
        if k:
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


fe, wr = per_family(load(fetch_dir), "FETCH_SIZE"), per_family(load(write_dir), "WRITE_SIZE")
nt = [k for k in fe if k.startswith(("gemm_nt", "gemm_nn", "gemm_pk", "gemm_p8"))]
launches = sum(fe[k][0] for k in nt)
fetch_kb = sum(fe[k][1] for k in nt) / launches
write_kb = sum(wr[k][1] for k in nt) / max(1, sum(wr[k][0] for k in nt))
traffic = {
    "kernel_family": "gemm_p8h_kernel / gemm_p8_kernel / gemm_pk_kernel / gemm_nt_* (bf16): every forward and dgrad projection",
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-extras --no-parity --no-roofline; launches of the last step only (no autotune trials)",
    "launches": launches, "fetch_kb_per_launch_raw": fetch_kb, "write_kb_per_launch": write_kb,
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted too",
    "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "per_family_kb_per_launch": {k: {"fetch_raw": fe[k][1] / fe[k][0], "write": wr[k][1] / max(1, wr[k][0]), "launches": fe[k][0]} for k in sorted(fe)},
}
json.dump(traffic, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
if sq_dir == "-":
    print(json.dumps({k: v for k, v in traffic.items() if k != "per_family_kb_per_launch"}, indent=1))
    sys.exit(0)
sq = load(sq_dir)
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES"]
agg = {n: per_family(sq, n) for n in names}
out = [f"# rocprofv3 --pmc (SQ / GRBM) on the bench step ({tag})", "",
       "command: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES",
       "--output-format csv -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-extras --no-parity --no-roofline --no-graph`",
       "(own pass, no trace domains; sums over the launches of the last step, i.e. without the autotune trials of the warm-up).", "",
       "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES.", "",
       "| kernel family | launches | MFMA busy cycles | GRBM_GUI_ACTIVE | MFMA utilisation | SQ_LDS_BANK_CONFLICT | conflict / LDS active |", "|---|---|---|---|---|---|---|"]
for k in sorted(agg["GRBM_GUI_ACTIVE"], key=lambda k: -agg["GRBM_GUI_ACTIVE"][k][1]):
    mf, gui = agg["SQ_VALU_MFMA_BUSY_CYCLES"][k][1], agg["GRBM_GUI_ACTIVE"][k][1]
    bc, la = agg["SQ_LDS_BANK_CONFLICT"][k][1], agg["SQ_LDS_IDX_ACTIVE"][k][1]
    out.append(f"| `{k}` | {agg['GRBM_GUI_ACTIVE'][k][0]} | {mf:.3g} | {gui:.3g} | {100 * mf / (gui / 8 * 1024):.1f} % | {bc:.3g} | {100 * bc / max(la, 1):.1f} % |")
open(os.path.join(root, "profiles", f"{tag}_pmc_sq.md"), "w").write("\n".join(out) + "\n")
print(json.dumps({k: v for k, v in traffic.items() if k != "per_family_kb_per_launch"}, indent=1))
print("\n".join(out[8:]))
