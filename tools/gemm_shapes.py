"""Round 6: every forward / dgrad GEMM shape of one bench step replayed on its own, the product's pick beside the
vendor library's kernel on the SAME shape (probe only - the product never calls the vendor library), with tile / round counts and
the per-shape floor max(MFMA time, HBM time).

  1. the shapes of a step:   VLNI_GEMM_SHAPES_OUT=gpurun_out/x/shapes.json python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-parity
  2. the replay, traced:     rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/gemm_shapes.py run shapes.json run.json
  3. the table:              python tools/gemm_shapes.py report run.json <dir> "title" > profiles/r06_gemm_shapes.md

`run`: per shape, a marker kernel (torch fill), N eager launches of the product's autotuned pick with the step's epilogue kind, a marker, N calls
of torch.matmul (closed by a third marker) on the concatenated rows (plain contraction, no epilogue: what the vendor kernel costs before bias / GELU / residual passes);
also HIP-event times of 20 launches replayed from one graph (both). `report`: rocprofv3's kernel durations between the markers (median of
the product's launches, total / calls for the vendor's, which may be several kernels per call)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_TF, HBM_TBS, CUS = 2500.0, 8.0, 256
NREP = 10


def tiles_of(variant, rows_list, N):
    """(tile rows, tile columns, workgroups per CU) of the pipelines the autotuner can pick (ops.GEMM_VARIANTS ids as passed to the C-ABI)."""
    geo = {15: (256, 256, 1), 32: (256, 128, 1), 14: (128, 128, 2), 7: (256, 256, 1), 6: (256, 128, 1), 8: (128, 256, 1), 12: (192, 128, 2), 13: (128, 192, 2),
           9: (64, 128, 4), 10: (128, 64, 4), 11: (64, 64, 4)}.get(variant, (128, 128, 2))
    tm, tn, per_cu = geo
    t = sum(-(-r // tm) for r in rows_list) * -(-N // tn)
    return tm, tn, per_cu, t


def run(shapes_path, out_path):
    import torch
    from vln_imagine_amd import ops
    dt = torch.bfloat16
    shapes = json.load(open(shapes_path))
    rnd = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).to(dt)
    marker = torch.empty(1 << 20, device="cuda")
    out = []

    def graph_us(fn, n=20):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * n) * 1e3

    for i, sh in enumerate(shapes):
        M, N, K, kind = sh["M"], sh["N"], sh["K"], sh["kind"]
        if M < 256:
            continue
        rows = [int(x) for x in kind.split("dual ")[1].split("+")] if "dual" in kind else [M]
        act = int(kind.split("act")[1].split()[0])
        dact = int(kind.split("dact")[1].split()[0])
        has_res, has_drop, has_pre = " res" in kind, " drop" in kind, " pre" in kind
        a = [rnd(r, K) for r in rows]
        w = [rnd(N, K, sc=0.05) for _ in rows]
        bias = [torch.randn(N, device="cuda") if not dact else None for _ in rows]
        res = [rnd(r, N) if has_res else None for r in rows]
        z = [rnd(r, N, sc=1.0) for r in rows]
        kw = dict(act=act, dact=dact)
        if len(rows) == 2:
            kw.update(bias=tuple(bias), residual=tuple(res), preact=tuple(z) if has_pre else (None, None), dact_src=tuple(z) if dact else (None, None))
            if has_drop:
                kw["drop"] = (0.1, (11, 12))
            ours = lambda: ops.gemm_nt2(tuple(a), tuple(w), **kw)
            key = (dt, rows[0], rows[1], N, K, act, dact, has_res, has_pre, False)
        else:
            kw.update(bias=bias[0], residual=res[0], preact=z[0] if has_pre else None, dact_src=z[0] if dact else None)
            if has_drop:
                kw["drop"] = (0.1, 11)
            ours = lambda: ops.gemm_nt(a[0], w[0], **kw)
            key = (dt, M, N, K, act, dact, has_res, has_pre, False)
        acat = torch.cat(a, 0)
        wt = w[0].t()
        vendor = lambda: torch.matmul(acat, wt)
        ours(); vendor()                       # autotune + warm
        ev_ours, ev_vendor = graph_us(ours), graph_us(vendor)
        torch.cuda.synchronize()
        marker.fill_(float(3 * i))
        for _ in range(NREP):
            ours()
        marker.fill_(float(3 * i + 1))
        for _ in range(NREP):
            vendor()
        marker.fill_(float(3 * i + 2))            # closes the vendor segment (what follows is the next shape's set-up and autotune)
        torch.cuda.synchronize()
        variant = ops._GEMM_BEST.get(key)
        out.append(dict(sh, rows=rows, variant=variant, event_us=round(ev_ours, 1), vendor_event_us=round(ev_vendor, 1)))
        print(f"{i:3d} M={M:6d} N={N:5d} K={K:5d} {kind:40s} v{variant}: {ev_ours:7.1f} us  vendor {ev_vendor:7.1f} us", flush=True)
        del a, w, res, z, acat
    marker.fill_(-1.0)
    torch.cuda.synchronize()
    json.dump(out, open(out_path, "w"), indent=1)


def report(run_path, trace_dir, title):
    rows = json.load(open(run_path))
    files = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    segs = []
    if files:
        ks = []
        for f in files:
            for r in csv.DictReader(open(f)):
                ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"],
                           (r["Grid_Size_X"], r["SGPR_Count"], r["VGPR_Count"])))
        ks.sort()
        fills = [k for k in ks if "FillFunctor" in k[2]]
        mark = fills[-1][3]                           # the marker tensor's fill kernel (the run ends with one); other fills (zeros, masks) have other grids
        cur = None
        for _, d, name, sig in ks:
            if "FillFunctor" in name and sig == mark:
                cur = []
                segs.append(cur)
            elif cur is not None:
                cur.append((d / 1e3, name))
        assert len(segs) == 3 * len(rows) + 1, (len(segs), len(rows))
    print(f"# {title}\n")
    print("One row per (shape, epilogue kind) of the instrumented step (`bench.py`, `roofline.bound_per_shape`), replayed on its own by `tools/gemm_shapes.py`. "
          "`us` = rocprofv3 kernel duration (median of 10 eager launches between marker kernels; `event` = HIP events around 20 launches replayed from one graph). "
          "`vendor` = `torch.matmul` on the concatenated rows: the vendor library's kernel(s) for the PLAIN contraction on the same shape - no bias, GELU, "
          "dropout or residual, which the product's launch includes (probe only). `floor` = max(flops / 2.5 PFLOP/s, bytes / 8 TB/s) with bytes = A + B + C + the "
          "epilogue's tensors; `tiles / rounds / fill` = output tiles of the picked pipeline, tiles / (256 CUs x workgroups per CU), and how full the last round is; "
          "`CUs` = CUs that get a tile when there is less than one round.\n")
    print("| rows | N | K | kind | launches / step | pick | us (rocprof) | event us | TFLOP/s | vendor us (rocprof) | vendor event us | ours / vendor | floor us (bound) | of floor | tiles | rounds | last-round fill | CUs |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot = [0.0, 0.0, 0.0, 0.0]
    for i, r in enumerate(rows):
        ours_us, ven_us = r["event_us"], r["vendor_event_us"]
        if len(segs) >= 3 * i + 3:
            mine = sorted(d for d, n in segs[3 * i] if "gemm" in n)
            if mine:
                ours_us = mine[len(mine) // 2]
            v = [d for d, n in segs[3 * i + 1] if "Fill" not in n]
            if v:
                ven_us = sum(v) / NREP
        fl = 2.0 * r["M"] * r["N"] * r["K"]
        floor = max(r["mfma_us"], r["hbm_us"])
        tm, tn, per_cu, t = tiles_of(r["variant"], r["rows"], r["N"])
        cap = CUS * per_cu
        rounds = t / cap
        last = (t % cap) / cap if t % cap else 1.0
        cus = min(CUS, -(-t // per_cu)) if t < cap else CUS
        n = r["launches"]
        tot[0] += n * ours_us; tot[1] += n * ven_us; tot[2] += n * floor; tot[3] += n * fl
        print(f"| {'+'.join(str(x) for x in r['rows'])} | {r['N']} | {r['K']} | {r['kind'].split(' dual')[0]} | {n} | v{r['variant']} {tm}x{tn} | {ours_us:.1f} | {r['event_us']:.1f} | "
              f"{fl / ours_us / 1e6:.0f} | {ven_us:.1f} | {r['vendor_event_us']:.1f} | {ours_us / ven_us:.2f} | {floor:.1f} ({r['bound']}) | {floor / ours_us:.2f} | {t} | {rounds:.2f} | {last:.2f} | {cus} |")
    print(f"\nWeighted by launches per step: product {tot[0] / 1e3:.2f} ms, vendor plain contraction {tot[1] / 1e3:.2f} ms, per-shape floor {tot[2] / 1e3:.2f} ms; "
          f"{tot[3] / tot[0] / 1e6:.0f} TFLOP/s = {tot[3] / tot[0] / 1e6 / PEAK_TF:.3f} of the nominal bf16 peak, {tot[2] / tot[0]:.3f} of the per-shape floor.")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        report(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "Forward / dgrad GEMM shapes of one step")
