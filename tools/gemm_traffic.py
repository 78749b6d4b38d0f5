"""Round 6: where the forward / dgrad GEMM family's fabric traffic comes from. Per kernel family the counters
(`rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`, profiles/r06_pmc_traffic.json: bytes per launch, FETCH doubled per the gfx950 correction) beside
a model built from the step's own shape list (tools/gemm_shapes.py `run` output: shape, epilogue kind, launches, picked pipeline):

    reads  = A + the epilogue's read operand (residual or GELU' source) + X * B        X = L2 slices that stream the weight panel
    writes = C + the stored pre-activation / GELU'

with X = 8 (every XCD's L2 pulls its own copy of the [N, K] weight panel: the tile order gives each XCD a contiguous range of ROW panels with
all their column tiles, so A, the residual and C are private to one L2) against X = 1 (what `algorithmic bytes` counts).
usage: python tools/gemm_traffic.py shapes_run.json pmc_traffic.json "title" > profiles/r06_gemm_traffic.md"""
import json
import sys

FAMILY = {15: "gemm_p8", 32: "gemm_p8h", 14: "gemm_pk"}
for v in (6, 7, 8, 9, 10, 11, 12, 13):
    FAMILY[v] = "gemm_nt_big"
for v in (0, 1, 2, 3, 4, 5, None):
    FAMILY[v] = "gemm_nt_glds"

rows = json.load(open(sys.argv[1]))
pmc = json.load(open(sys.argv[2]))
title = sys.argv[3] if len(sys.argv) > 3 else "Fabric traffic of the forward / dgrad GEMM family (r06)"
fam = {}
for r in rows:
    M, N, K, kind, n = r["M"], r["N"], r["K"], r["kind"], r["launches"]
    a, b, c = 2.0 * M * K, 2.0 * N * K, 2.0 * M * N
    epi_r = c if (" res" in kind or "dact3" in kind or "dact1" in kind) else 0.0
    epi_w = c if " pre" in kind else 0.0
    f = fam.setdefault(FAMILY.get(r.get("variant"), "gemm_nt_glds"), dict(n=0, a=0.0, b=0.0, c=0.0, er=0.0, ew=0.0))
    f["n"] += n; f["a"] += n * a; f["b"] += n * b; f["c"] += n * c; f["er"] += n * epi_r; f["ew"] += n * epi_w
print(f"# {title}\n")
print("Counters: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes of `bench.py --steps 1 --warmup 2 --no-graph`, launches of the last step; FETCH_SIZE doubled: "
      "gfx950 counts 64 B per 128-B request on wide coalesced reads, `profiles/r01_fetch_calibration.md`; Infinity-Cache hits are counted, so this is FABRIC traffic of the L2s, not HBM traffic). "
      "Model: reads = A + epilogue read operand + X x B, writes = C + stored pre-activation, per launch, averaged over the family's launches of one step (`tools/gemm_traffic.py`; the launch counts "
      "per family follow the shape table's picks, the counter pass's own autotune may pick another pipeline for a few shapes). MB per launch.\n")
print("| kernel family | launches (model / counters) | A | epilogue read | B | C + stored | model reads X = 1 | model reads X = 8 | measured reads | model writes | measured writes | measured / algorithmic (X = 1) | measured / model (X = 8) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
tot = dict(m1=0.0, m8=0.0, mr=0.0, w=0.0, mw=0.0, n=0)
for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["a"]):
    p = pmc["per_family_kb_per_launch"].get(name)
    if not p or not f["n"]:
        continue
    n = f["n"]
    A, B, C, ER, EW = (f[k] / n / 1e6 for k in ("a", "b", "c", "er", "ew"))
    r1, r8, w = A + ER + B, A + ER + 8 * B, C + EW
    mr, mw = 2.0 * p["fetch_raw"] * 1024 / 1e6, p["write"] * 1024 / 1e6
    print(f"| `{name}` | {n} / {p['launches']} | {A:.1f} | {ER:.1f} | {B:.2f} | {C + EW:.1f} | {r1:.1f} | {r8:.1f} | {mr:.1f} | {w:.1f} | {mw:.1f} | {(mr + mw) / (r1 + w):.2f} | {(mr + mw) / (r8 + w):.2f} |")
    k = p["launches"]
    tot["m1"] += k * (r1 + w); tot["m8"] += k * (r8 + w); tot["mr"] += k * (mr + mw); tot["n"] += k
print(f"\nWhole family ({tot['n']} launches): measured {tot['mr'] / tot['n']:.1f} MB per launch against {tot['m1'] / tot['n']:.1f} algorithmic (one copy of every operand, epilogue operands included) = "
      f"{tot['mr'] / tot['m1']:.2f} x, and against {tot['m8'] / tot['n']:.1f} with the weight panel fetched once per XCD = {tot['mr'] / tot['m8']:.2f} x. `roofline.traffic_ratio` in the bench line divides by A + B + C "
      "alone (no epilogue operands), which is why it reads higher.")
