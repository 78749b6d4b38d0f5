#!/usr/bin/env python3
"""bench.py -- episodes/sec (fwd+bwd) of the HAMT-Imagine hot path on MI355X (BASELINE.json metric).

One "step" = one batch of synthetic R2R episodes through the HIP path:
  1 'language' + 1 'imagine' + 1 'align_with_contrastive_loss' + 1 'history'(CLS)
  + T x ('visual' + 'history' step) + CE(sum) per step -> loss = ml*0.2/B + 0.5*aux
  -> backward -> [RCCL gradient all-reduce if N > 1] -> clip_grad_norm(40) -> AdamW.
Default workload = BASELINE.json configs[1]: 9 L + 4 X + 2 hist-pano layers, batch 64 per GPU,
80 text tokens, 37 observation tokens, 6 imaginations, T = 6, bf16 compute, all layers trainable.
Inputs are resident in HBM before the timed region. N > 1: one process per GPU (torchrun), weak scaling
(64 episodes per GPU), value = all ranks' episodes / max-over-ranks time.

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel = the MFMA GEMM, timed with HIP events
in an instrumented pass after the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_F32_PEAK_TFLOPS = 157.3


def episode_flops(cfg, B, L, V, I, T, shipped_freeze):
    """Algorithmic FLOPs of one episode batch (SURVEY.md section 8d formulas; multiply-add = 2, bwd = 2x fwd)."""
    H, FF = 768, 3072
    bert = lambda S: 24 * S * H * H + 4 * S * S * H
    xl = lambda Lt, Lv: 32 * (Lt + Lv) * H * H + 8 * Lt * Lv * H + 4 * (Lt * Lt + Lv * Lv) * H
    lang = cfg.num_l_layers * bert(L)
    hist_step = 2 * H * H + cfg.num_h_pano_layers * bert(36) + 2 * 36 * H * H
    aux = 2 * I * (768 * 512 + 512 * 512 + 512 * 768)
    total = 0.0
    total += lang * (1 if shipped_freeze else 3)
    total += aux * 3
    for t in range(T):
        Lv = (1 + t) + V
        visual = cfg.num_x_layers * xl(L + I, Lv) + 2 * V * H * H * 2   # + obs embed + head
        total += visual * 3
        total += hist_step * (1 if shipped_freeze else 3)
    return total * B


def make_model(cfg, dtype, device):
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
    from vln_imagine_amd.hamt.spec import param_shapes
    m = NavCMT(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
    return m.to(device).eval().set_compute_dtype(dtype)      # eval(): dropout p = 0 as in the survey probe


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("VLNI_CPU_THREADS", "64"))))


def cpu_baseline(cfg, args):
    """The CPU oracle (plain PyTorch fp32 restatement pinned to the reference's golden vectors) on the host cores."""
    from oracle.hamt_oracle import HamtOracle
    from vln_imagine_amd import synth
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
    from vln_imagine_amd.hamt.spec import param_shapes
    cores = host_cores()
    log(f"cpu_baseline: oracle on {cores} host threads, batch {args.cpu_batch}")
    torch.set_num_threads(cores)
    Bc = args.cpu_batch
    ep = synth.HamtEpisode(tag="cpu", B=Bc, L=args.L, V=args.V, I=args.I, T=args.T, ragged=False)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-5)
    et = EpisodeTensors(ep, "cpu")
    model = HamtOracle(cfg, sd)
    t0 = time.time()
    out = run_episode(model, et, keep=False)
    out["loss"].backward()
    torch.nn.utils.clip_grad_norm_(list(sd.values()), 40.0)
    opt.step()
    dt = time.time() - t0
    return {"value": Bc / dt, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"1 step of {Bc} episodes (same model/T/shapes as the GPU workload, fp32, torch {torch.__version__} CPU, "
                      f"fwd+bwd+clip+AdamW), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--T", type=int, default=6)
    ap.add_argument("--L", type=int, default=80)
    ap.add_argument("--V", type=int, default=37)
    ap.add_argument("--I", type=int, default=6)
    ap.add_argument("--freeze", default="none", choices=["none", "shipped"])
    ap.add_argument("--time-batched", action="store_true",
                    help="HAMT: run the T teacher-forced steps as one [T*B] batch (same results, SURVEY 8f rank 1)")
    ap.add_argument("--no-time-batched-extra", action="store_true")
    ap.add_argument("--grad-comm", default="bf16", choices=["bf16", "fp32"],
                    help="dtype of the gradient all-reduce payload for N > 1 (arena stays fp32)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True,
                    help="replay the step from captured hipGraphs (default)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel from Python")
    ap.add_argument("--model", default="hamt", choices=["hamt", "duet"],
                    help="hamt = BASELINE.json configs[1] (the metric's config); duet = configs[3] (batch 32)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("VLNI_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" IS RCCL on ROCm; VLNI_DIST_BACKEND=gloo + VLNI_ONE_GPU=1 rehearse the multi-rank path on a 1-GPU box
        backend = os.environ.get("VLNI_DIST_BACKEND", "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}       # bind the communicator to this rank's GPU up front
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from vln_imagine_amd import ops, synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
    from vln_imagine_amd.train import FlatTrainer

    shipped = args.freeze == "shipped"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.model == "duet":
        from vln_imagine_amd.duet.config import DuetConfig
        from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode as duet_run
        from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
        from vln_imagine_amd.duet.spec import param_shapes as duet_shapes
        if args.batch == 64:
            args.batch = 32                                  # BASELINE.json configs[3]
        cfg = DuetConfig(fix_lang_embedding=shipped, update_lang_bert=not shipped)
        model = GlocalTextPathNavCMT(cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(duet_shapes(cfg).items()).items()})
        model = model.to(dev).eval().set_compute_dtype(dtype)
        ep = synth.DuetEpisode(tag=f"bench{rank}", B=args.batch, L=args.L, V=36, I=args.I, T=args.T, ragged=False)
        et = DuetEpisodeTensors(ep, dev)
        run_episode = duet_run
    else:
        cfg = HamtConfig(fix_lang_embedding=shipped, fix_hist_embedding=shipped, update_lang_bert=not shipped)
        model = make_model(cfg, dtype, dev)
        ep = synth.HamtEpisode(tag=f"bench{rank}", B=args.batch, L=args.L, V=args.V, I=args.I, T=args.T, ragged=False)
        et = EpisodeTensors(ep, dev)
        if args.time_batched:
            from vln_imagine_amd.hamt.episode import run_episode_time_batched
            run_episode = lambda m, e, criterion=None, keep=False: run_episode_time_batched(m, e, criterion=criterion)
    trainer = FlatTrainer(model, lr=1e-5, grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)

    def step():
        trainer.zero_grad()
        out = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)
        out["loss"].backward()
        trainer.allreduce_grads()
        trainer.step()
        return out["loss"]

    eager_step = step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model + episode ready on {dev}; warmup {args.warmup}, steps {args.steps}, dtype {args.dtype}")
    eager_s = float("inf")
    for i in range(args.warmup):
        tw = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        eager_s = min(eager_s, time.perf_counter() - tw)
        log(f"warmup {i}: {1e3 * (time.perf_counter() - tw):.1f} ms loss {float(loss.detach()):.5f}")
    launch = "eager (one kernel launch per op from Python)"
    if args.graph:
        # same step, replayed from two captured hipGraphs (fwd+bwd+wgrad flush | clip+AdamW) with the RCCL all-reduce between
        def fwd_bwd():
            out = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)
            out["loss"].backward()
            return out["loss"]
        loss = None                                 # drop the last eager autograd graph (its AccumulateGrad nodes) before capturing
        import gc
        gc.collect()
        slow, captured = True, None
        try:
            captured = trainer.capture(fwd_bwd, warmup=1)
            loss = captured()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(2):
                loss = captured()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg) / 2
            log(f"step captured into hipGraphs; loss {float(loss.detach()):.5f}; replay {1e3 * tg:.1f} ms vs eager {1e3 * eager_s:.1f} ms")
            slow = tg > 1.3 * eager_s                 # never seen on a dedicated GPU; two processes SHARING one GPU replay pathologically slowly
        except Exception as e:                      # keep measuring: fall back to the eager step and say so
            log(f"graph capture failed ({type(e).__name__}: {e})")
        if world > 1:                                 # one decision for the whole job (every rank reaches this all-reduce)
            flag = torch.tensor([1.0 if slow else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            slow = bool(flag.item() > 0)
        if slow:
            log("no usable graph replay on this box: timing the eager step")
            step = eager_step
        else:
            step = captured
            launch = "hipGraph replay (fwd+bwd+wgrad | clip+AdamW), all-reduce eager between"
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    log(f"timed: {ms:.2f} ms/step")
    eps = args.batch * world / (dt / args.steps)
    flops = episode_flops(cfg, args.batch, args.L, args.V, args.I, args.T, shipped) if args.model == "hamt" \
        else 150e9 * args.batch * args.T / 6.0          # DUET episode T=6 all-trainable ~150 GF/sample
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS

    roof = None
    if not args.no_roofline:           # EVERY rank runs the instrumented step (it contains the gradient all-reduce)
        # instrumented pass: HIP events (torch.cuda.Event on the launch stream = torch's current stream) around
        # every vlni_gemm_nt launch of ONE more step; not part of the timed region above.
        rec, epi = [], [0.0]
        orig = ops.gemm_nt

        def _epi_bytes(k):              # tensors the fused epilogue reads / writes besides C: residual, GELU' source, pre-activation
            n = 0
            for key in ("residual", "dact_src", "preact"):
                v = k.get(key)
                for t_ in (v if isinstance(v, (tuple, list)) else (v,)):
                    if torch.is_tensor(t_):
                        n += t_.numel() * t_.element_size()
            return n

        def timed(a, b, *p, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, *p, **k)
            e1.record()
            n_ = b.n if isinstance(b, ops.WT) else b.t.shape[1] if isinstance(b, ops.KN) else b.shape[0]      # WT / KN: dgrad operand handles
            rec.append((2.0 * a.shape[0] * n_ * a.shape[1], e0, e1, (a.shape[0], n_, a.shape[1])))
            epi[0] += _epi_bytes(k)
            return r

        orig2 = ops.gemm_nt2

        def timed2(a, b, *p, **k):                     # dual-problem launches (language + vision stream in one launch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig2(a, b, *p, **k)
            e1.record()
            rows = a[0].shape[0] + a[1].shape[0]
            n_ = b[0].n if isinstance(b[0], ops.WT) else b[0].t.shape[1] if isinstance(b[0], ops.KN) else b[0].shape[0]
            rec.append((2.0 * rows * n_ * a[0].shape[1], e0, e1, (rows, n_, a[0].shape[1])))
            epi[0] += _epi_bytes(k)
            return r

        step = eager_step
        ops.gemm_nt, ops.gemm_nt2 = timed, timed2
        try:
            # keep the stream busy while the host enqueues the step, so that each event pair brackets the kernel alone and not
            # the host's gap between recording the event and launching (otherwise the average reads ~35 % above rocprof's)
            torch.cuda._sleep(int(0.05 * getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2.4e6) * 1e3))
            step()
            torch.cuda.synchronize()
        finally:
            ops.gemm_nt, ops.gemm_nt2 = orig, orig2
        log("instrumented roofline step done")
        tot_f = sum(r[0] for r in rec)
        tot_ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        if os.environ.get("VLNI_GEMM_BREAKDOWN"):
            agg = {}
            for f, e0, e1, shp in rec:
                a_ = agg.setdefault(shp, [0, 0.0, 0.0])
                a_[0] += 1; a_[1] += e0.elapsed_time(e1); a_[2] += f
            for shp, (n, ms_, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                log(f"gemm M={shp[0]:6d} N={shp[1]:5d} K={shp[2]:5d}: {n:4d} calls {ms_:7.2f} ms {f / ms_ / 1e9:7.1f} TF/s")
        ach = tot_f / (tot_ms * 1e-3) / 1e12
        traffic = None
        try:      # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/); never live
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
            if pmc and args.dtype == "bf16" and args.model == "hamt" and not args.time_batched:      # collected on the default workload only
                traffic = round(json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))["bytes_per_launch"])
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "kernel": "gemm_nt_* / gemm_nn_glds_kernel <%s>" % ("__bf16" if args.dtype == "bf16" else "float"),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(sum(2.0 * (m_ * k_ + n_ * k_ + m_ * n_) for _, _, _, (m_, n_, k_) in rec) / len(rec)),
                "algorithmic_bytes_per_launch_with_epilogue": round((sum(2.0 * (m_ * k_ + n_ * k_ + m_ * n_) for _, _, _, (m_, n_, k_) in rec) + epi[0]) / len(rec)),
                "launches_per_step": len(rec), "avg_launch_us": round(tot_ms * 1e3 / len(rec), 2),
                "avg_gflop_per_launch": round(tot_f / len(rec) / 1e9, 3),
                "gemm_share_of_step": round(tot_ms / ms, 3),
                "step_algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                "step_frac_of_peak": round(flops / (ms * 1e-3) / 1e12 / peak, 4)}

    tb = None
    if args.model == "hamt" and not args.time_batched and not args.no_time_batched_extra:
        # extra, reported beside `value` (never instead of it): the same episodes with the T teacher-forced steps run as one
        # [T*B] batch (identical logits/loss/gradients, tests/test_hamt_gpu.py)
        from vln_imagine_amd.hamt.episode import run_episode_time_batched

        def step_tb():
            trainer.zero_grad()
            out = run_episode_time_batched(model, et, criterion=ops.cross_entropy_sum)
            out["loss"].backward()
            trainer.allreduce_grads()
            trainer.step()
            return out["loss"]

        eager_tb, eager_tb_s = step_tb, float("inf")
        for _ in range(max(2, args.warmup)):
            tw = time.perf_counter()
            step_tb()
            torch.cuda.synchronize()
            eager_tb_s = min(eager_tb_s, time.perf_counter() - tw)
        if args.graph and launch.startswith("hipGraph"):
            import gc
            gc.collect()

            def fwd_bwd_tb():
                out = run_episode_time_batched(model, et, criterion=ops.cross_entropy_sum)
                out["loss"].backward()
                return out["loss"]
            slow, cap_tb = True, None
            try:
                cap_tb = trainer.capture(fwd_bwd_tb, warmup=1)
                cap_tb()
                torch.cuda.synchronize()
                tw = time.perf_counter()
                cap_tb()
                torch.cuda.synchronize()
                slow = time.perf_counter() - tw > 1.3 * eager_tb_s
            except Exception as e:
                log(f"time-batched graph capture failed ({type(e).__name__}: {e}); eager")
            if world > 1:
                flag = torch.tensor([1.0 if slow else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                slow = bool(flag.item() > 0)
            step_tb = eager_tb if slow else cap_tb
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_tb()
        fence()
        dtb = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dtb], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtb = float(tt.item())
        tb = {"value": round(args.batch * world / (dtb / args.steps), 2), "unit": "episodes/s",
              "ms_per_step": round(dtb / args.steps * 1e3, 3),
              "step_algorithmic_tflops": round(flops / (dtb / args.steps) / 1e12, 2),
              "note": "T steps as one [T*B] batch under teacher forcing; same results as the step-by-step calls"}
        log(f"time-batched extra: {tb['ms_per_step']} ms/step")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "hamt":
        cpu = cpu_baseline(cfg, args)

    if rank == 0:
        line = {
            "metric": "episodes/sec (fwd+bwd) HAMT-Imagine 9L, batch 64" if args.model == "hamt"
            else "episodes/sec (fwd+bwd) DUET-Imagine 9L+2pano+4+4X, batch 32", "value": round(eps, 2), "unit": "episodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{'HAMT-Imagine 9L+4X+2pano' if args.model == 'hamt' else 'DUET-Imagine 9L+2pano+4global+4local X, map 5+3t nodes'}, batch {args.batch}/GPU, {args.L} text, {args.V} obs tokens, "
                                   f"{args.I} imaginations, T={args.T} steps/episode, freeze={args.freeze}, dropout p=0"
                                   + (", steps time-batched (teacher forcing)" if args.time_batched else ", step-by-step calls"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "launch": launch,
                       "grad_allreduce": (args.grad_comm + " payload, flat arena, 128-MiB chunks, RCCL") if world > 1 else "none (1 GPU)",
                       "steps_per_sec": round(args.T * args.batch * world / (dt / args.steps), 1),
                       "loss": round(float(loss), 5)},
            "roofline": roof, "cpu_baseline": cpu, "time_batched": tb,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
