#!/usr/bin/env python3
"""bench.py -- episodes/sec (fwd+bwd) of the HAMT-Imagine hot path on MI355X (BASELINE.json metric).

One "step" = one batch of synthetic R2R episodes through the HIP path:
  1 'language' + 1 'imagine' + 1 'align_with_contrastive_loss' + 1 'history'(CLS)
  + T x ('visual' + 'history' step) + CE(sum) per step -> loss = ml*0.2/B + 0.5*aux
  -> backward -> [RCCL gradient all-reduce if N > 1] -> clip_grad_norm(40) -> AdamW.
Default workload = BASELINE.json configs[1]: 9 L + 4 X + 2 hist-pano layers, batch 64 per GPU,
80 text tokens, 37 observation tokens, 6 imaginations, T = 6, bf16 compute, all layers trainable.
Inputs are resident in HBM before the timed region. Weak scaling (64 episodes per GPU), value = all ranks' episodes /
max-over-ranks time.

N > 1: one process per GPU over RCCL. Under torchrun the ranks come from the environment; `python bench.py --gpus N` alone
starts the N rank processes itself - from a parent that never touches the GPU - and relays rank 0's line
(reference launch: VLN-HAMT/finetune_src/utils/distributed.py:13-71, DDP wrap r2r/agent_cmt.py:61-63).

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel = the MFMA GEMM family, timed with HIP events on the
launch stream in an instrumented pass after the timed region), `cpu_baseline` (the CPU oracle on a bounded sample, N = 1),
`bf16_vs_fp32` (error of the timed bf16 path against the fp32 parity path of the same model on a B = 8 slice) and, at N = 1,
`extras`: the same episodes through the other two drivers (one autograd graph per call; time-batched), in eval mode (dropout off), with the
shipped freeze, at T = 1, in fp16 and in fp32. Default = the reference's training mode (model.train(), dropout p = 0.1 inside the kernels).
"""
import argparse
import json
import socket
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402   (importing torch does not initialise the GPU)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0              # HBM3E, same guide ("HBM ~8 TB/s"; ~6.3 TB/s is what a streaming kernel reaches)
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_F32_PEAK_TFLOPS = 157.3


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# =====================================================================================================================
#  N ranks from one command line
# =====================================================================================================================
def comm_self_check(dev, world, rank, local, device_name):
    """What the COMMUNICATOR says about the job, measured through it: the sum of one 1 per rank (= how many ranks the collective really spans),
    each rank's (rank, local rank, device) as gathered over it, and how many distinct devices those are."""
    one = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(one)
    got = [None] * world
    dist.all_gather_object(got, (rank, local, device_name))
    return {"comm_size": int(round(float(one.item()))), "comm_ranks": sorted(got), "distinct_devices": len({g[2] for g in got})}


def gather_floats(x, world, dev):
    t = torch.tensor([float(x)], device=dev, dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [round(float(o.item()), 3) for o in out]


def launch_ranks(n):
    """Starts n copies of this command line as rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), waits, and returns
    the worst exit code. This parent makes no GPU call (a process that holds the device must not start others on these boxes);
    rank 0 inherits stdout, so its JSON line is this command's JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                c = p.poll()
                if c is None:
                    continue
                procs.remove(p)
                rc = max(rc, abs(c))
                if c != 0:                         # one rank died: the others would wait in a collective forever
                    for q in procs:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    return rc


# =====================================================================================================================
#  workloads
# =====================================================================================================================
def episode_flops(cfg, B, L, V, I, T, shipped_freeze):
    """Algorithmic FLOPs of one HAMT episode batch (SURVEY.md section 8d formulas; multiply-add = 2, bwd = 2x fwd)."""
    H = 768
    bert = lambda S: 24 * S * H * H + 4 * S * S * H
    xl = lambda Lt, Lv: 32 * (Lt + Lv) * H * H + 8 * Lt * Lv * H + 4 * (Lt * Lt + Lv * Lv) * H
    lang = cfg.num_l_layers * bert(L)
    hist_step = 2 * H * H + cfg.num_h_pano_layers * bert(36) + 2 * 36 * H * H
    aux = 2 * I * (768 * 512 + 512 * 512 + 512 * 768)
    total = lang * (1 if shipped_freeze else 3) + aux * 3
    for t in range(T):
        Lv = (1 + t) + V
        visual = cfg.num_x_layers * xl(L + I, Lv) + 2 * V * H * H * 2   # + obs embed + head
        total += visual * 3 + hist_step * (1 if shipped_freeze else 3)
    return total * B


class Workload:
    """Model + resident synthetic episode + the episode driver, for the product (device) or the CPU oracle."""

    def __init__(self, family, args, shipped, device, dtype=None, batch=None, T=None, tag="bench", oracle=False, model=None):
        from vln_imagine_amd import synth
        self.family, self.shipped = family, shipped
        B, T = batch or args.batch, T or args.T
        self.B, self.T = B, T
        if family == "duet":
            from vln_imagine_amd.duet.config import DuetConfig
            from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode
            from vln_imagine_amd.duet.spec import param_shapes
            self.cfg = DuetConfig(fix_lang_embedding=shipped, update_lang_bert=not shipped)
            ep = synth.DuetEpisode(tag=tag, B=B, L=args.L, V=36, I=args.I, T=T, ragged=False)
            self.et, self.logits_key = DuetEpisodeTensors(ep, device), "fused"
            self._run = run_episode
            self.flops = 150e9 * B * T / 6.0                     # DUET episode T=6 all-trainable ~150 GF/sample
            self.label = "DUET-Imagine 9L+2pano+4global+4local X, map 5+3t nodes"
        else:
            from vln_imagine_amd.hamt.config import HamtConfig
            from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
            from vln_imagine_amd.hamt.spec import param_shapes
            self.cfg = HamtConfig(fix_lang_embedding=shipped, fix_hist_embedding=shipped, update_lang_bert=not shipped)
            ep = synth.HamtEpisode(tag=tag, B=B, L=args.L, V=args.V, I=args.I, T=T, ragged=False)
            self.et, self.logits_key = EpisodeTensors(ep, device), "logits"
            self._run = run_episode
            self.flops = episode_flops(self.cfg, B, args.L, args.V, args.I, T, shipped)
            self.label = "HAMT-Imagine 9L+4X+2pano"
        self._shapes, self._weights = param_shapes(self.cfg), None
        if oracle:
            self.sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in self.weights.items()}
            if family == "duet":
                from oracle.duet_oracle import DuetOracle as Oracle
            else:
                from oracle.hamt_oracle import HamtOracle as Oracle
            self.model = Oracle(self.cfg, self.sd)
        else:
            self.model = model if model is not None else self.build(device, dtype)      # model: another episode for a resident model

    @property
    def weights(self):
        """Closed-form synthetic weights (a hash of parameter name + index, vln_imagine_amd/synth.py): no checkpoint is shipped."""
        if self._weights is None:
            from vln_imagine_amd import synth
            self._weights = synth.fill_state_dict(self._shapes.items())
        return self._weights

    def build(self, device, dtype):
        if self.family == "duet":
            from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT as Net
        else:
            from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT as Net
        m = Net(self.cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in self.weights.items()})
        return m.to(device).eval().set_compute_dtype(dtype)      # eval(): dropout p = 0 (the survey's CPU probe); `train_mode` extra: p = 0.1

    def run(self, criterion=None, keep=False, model=None, et=None, mode="stepwise"):
        """mode: 'stepwise' = T autograd graphs, one per call (the reference agent's own pattern); 'taped' (HAMT) = the same
        step-by-step forward calls recorded on an episode tape + ONE episode-batched backward (valid for sampled rollouts too);
        'time_batched' (HAMT) = forward AND backward on T x B samples (teacher forcing only)."""
        kw = {} if criterion is None else {"criterion": criterion}
        if mode == "time_batched":
            if self.family == "duet":
                from vln_imagine_amd.duet.episode import run_episode_time_batched
            else:
                from vln_imagine_amd.hamt.episode import run_episode_time_batched
            return run_episode_time_batched(model or self.model, et or self.et, **kw)
        if mode == "taped":
            from vln_imagine_amd import ops
            if self.family == "duet":
                from vln_imagine_amd.duet.episode import run_episode_taped
            else:
                from vln_imagine_amd.hamt.episode import run_episode_taped
            if getattr(self, "tape", None) is None:
                self.tape = ops.EpisodeTape(self.T)
            return run_episode_taped(model or self.model, et or self.et, tape=self.tape, **kw)
        return self._run(model or self.model, et or self.et, keep=keep, **kw)


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


def cpu_baseline(args, shipped):
    """The CPU oracle (plain PyTorch fp32 restatement pinned to the reference's golden vectors) on the host cores: one warm-up
    step at a small batch (thread pools, allocator), then ONE timed step of the GPU workload's batch on all cores, and one at a
    quarter of the batch on 8 threads (the survey container's core count)."""
    cores = host_cores()

    def one(B, threads):
        torch.set_num_threads(threads)
        w = Workload(args.model, args, shipped, "cpu", batch=B, tag="cpu", oracle=True)
        params = list(w.sd.values())
        opt = torch.optim.AdamW(params, lr=1e-5)
        t0 = time.time()
        w.run()["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 40.0)
        opt.step()
        return time.time() - t0

    log(f"cpu_baseline: oracle on {cores} host threads (warm-up at batch 4, then batch {args.cpu_batch})")
    one(4, cores)
    dt = one(args.cpu_batch, cores)
    res = {"value": args.cpu_batch / dt, "unit": "episodes/s", "cores": cores, "kind": "port",
           "sample": f"1 step of {args.cpu_batch} episodes after a warm-up step at batch 4 (same model/T/shapes as the GPU workload, fp32, "
                     f"torch {torch.__version__} CPU, fwd+bwd+clip+AdamW), {dt:.1f} s wall"}
    if cores > 8 and not args.quick_cpu:
        b8 = max(4, args.cpu_batch // 4)
        d8 = one(b8, 8)
        res["threads8"] = {"value": b8 / d8, "unit": "episodes/s", "cores": 8, "sample": f"1 step of {b8} episodes, {d8:.1f} s wall"}
    torch.set_num_threads(cores)
    return res


# =====================================================================================================================
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY 8d: >= 50 after 10 warm-up; a step is ~30 ms)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="compute dtype; fp16 = BASELINE.json configs[4] (float16 MFMA + dynamic loss scaling, float32 masters)")
    ap.add_argument("--batch", type=int, default=None, help="episodes per GPU (default: 64 HAMT, 32 DUET = BASELINE.json configs[1] / [3])")
    ap.add_argument("--T", type=int, default=6)
    ap.add_argument("--L", type=int, default=80)
    ap.add_argument("--V", type=int, default=37)
    ap.add_argument("--I", type=int, default=6)
    ap.add_argument("--freeze", default="none", choices=["none", "shipped"])
    ap.add_argument("--lang-rows", default="all", choices=["all", "cls"],
                    help="HAMT: 'all' = every language row through the last cross-modal layer like the reference's NavCMT (the headline); "
                         "'cls' = only the row the agent reads (what the VLNBertCMT wrapper selects; identical results)")
    ap.add_argument("--mode", default=None, choices=["taped", "stepwise", "time_batched"],
                    help="HAMT episode driver (default taped): taped = step-by-step forward calls + ONE episode-batched backward (episode tape; "
                         "what a sampled rollout can use); stepwise = one autograd graph per call; time_batched = forward and backward on "
                         "T x B samples (teacher forcing only; HAMT). DUET: taped (maps padded to the episode's largest) or stepwise.")
    ap.add_argument("--time-batched", action="store_true", help="same as --mode time_batched")
    ap.add_argument("--train-mode", dest="train_mode", action="store_true", default=True,
                    help="model.train(): in-kernel dropout p = 0.1 - the reference's training mode (default)")
    ap.add_argument("--eval-mode", dest="train_mode", action="store_false", help="model.eval(): dropout p = 0 (rounds 1-2 quoted this)")
    ap.add_argument("--grad-comm", default="bf16", choices=["bf16", "fp32"],
                    help="dtype of the gradient all-reduce payload for N > 1 (arena stays fp32)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True,
                    help="replay the step from captured hipGraphs (default)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel from Python")
    ap.add_argument("--model", default="hamt", choices=["hamt", "duet"],
                    help="hamt = BASELINE.json configs[1] (the metric's config); duet = configs[3] (batch 32)")
    ap.add_argument("--cpu-batch", type=int, default=None, help="episodes of the CPU baseline step (default: the GPU batch)")
    ap.add_argument("--quick-cpu", action="store_true", help="skip the 8-thread CPU line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip time-batched / train-mode / shipped-freeze / T=1 / fp32 extra lines")
    ap.add_argument("--no-parity", action="store_true", help="skip the bf16-vs-fp32 error report")
    ap.add_argument("--dump-tune", default=None, help="write the kernel choices of this run (GEMM pipeline per shape, weight-gradient variant x split) to a file")
    ap.add_argument("--load-tune", default=None, help="start from the kernel choices of another run (--dump-tune): the counter passes (--no-graph) then "
                                                      "launch the very kernels the timed, graph-replayed run used")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 32 if args.model == "duet" else 64
    if args.cpu_batch is None:
        args.cpu_batch = args.batch
    if args.time_batched:
        args.mode = "time_batched"
    if args.model == "duet" and args.mode == "time_batched":
        raise SystemExit("bench.py: DUET has no time-batched driver (its map grows with the agent's moves); use --mode taped or stepwise")
    if args.mode is None:
        args.mode = "taped"
    args.time_batched = args.mode == "time_batched"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU (torchrun) or drop WORLD_SIZE")
    if os.environ.get("VLNI_BENCH_DRY_RUN"):          # CPU test of the launcher: ranks rendezvous over gloo and report, no GPU work
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local))
        chk = comm_self_check(torch.device("cpu"), world, rank, local, f"cpu:{local}")
        per_rank = gather_floats(0.0, world, torch.device("cpu"))
        if rank == 0:
            # the keys of the real line's config.rccl (tests/test_host_cpu.py asserts them): the first `--gpus 8` run checks itself
            rccl = dict(chk, backend="gloo (dry run)", world_size=dist.get_world_size(), per_rank_ms_per_step=per_rank, allreduce_ms_exposed=None,
                        ms_per_step_without_exchange=None, payload_bytes_per_step=None, payload_dtype=args.grad_comm, exchange=None)
            print(json.dumps({"n_gpus": world, "ranks": seen, "config": {"rccl": rccl}}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("VLNI_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl = None
    forced = world == 1 and os.environ.get("VLNI_FORCE_COLLECTIVES") == "1"      # one rank, but the whole exchange pipeline through RCCL
    if forced:
        os.environ.setdefault("MASTER_PORT", "29517")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" IS RCCL on ROCm; VLNI_DIST_BACKEND=gloo + VLNI_ONE_GPU=1 rehearse the multi-rank path on a 1-GPU box
        backend = os.environ.get("VLNI_DIST_BACKEND", "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}       # bind the communicator to this rank's GPU up front
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries rank 0's ONE JSON line, so the
        # communicator is brought up (init + a first collective) with fd 1 pointing at stderr
        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd, 1)
            os.close(keep_fd)
        props = torch.cuda.get_device_properties(dev)
        dev_name = f"{socket.gethostname()}/{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', local)}/{local}"
        rccl = {"backend": backend, "world_size": dist.get_world_size(),
                "forced_single_rank": True if forced else None}
        rccl.update(comm_self_check(dev, dist.get_world_size(), rank, local, dev_name))
        if rccl["comm_size"] != world or rccl["distinct_devices"] != (1 if (forced or os.environ.get("VLNI_ONE_GPU")) else world):
            raise SystemExit(f"bench.py: the communicator spans {rccl['comm_size']} ranks on {rccl['distinct_devices']} devices, --gpus says {world}")

    from vln_imagine_amd import ops
    from vln_imagine_amd.compare import compare_runs
    from vln_imagine_amd.train import FlatTrainer

    shipped = args.freeze == "shipped"
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    peak = MFMA_F32_PEAK_TFLOPS if args.dtype == "fp32" else MFMA_BF16_PEAK_TFLOPS            # f16 and bf16 MFMA run at the same rate
    scaler = dict(loss_scale=16384.0, growth_interval=2000) if args.dtype == "fp16" else {}      # GradScaler semantics (train_r2r.py:201-234)
    comm = torch.bfloat16 if args.grad_comm == "bf16" else None

    def fence():
        torch.cuda.synchronize()
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(flag):
        """One decision for the whole job (every rank reaches this all-reduce)."""
        if world > 1:
            f = torch.tensor([1.0 if flag else 0.0], device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            return bool(f.item() > 0)
        return flag

    median_ms = {}

    def measure(w, trainer, steps, warmup, mode="stepwise", graph=True, what="step"):
        """W untimed warm-up steps, capture, then EXACTLY `steps` steps between fences; max over ranks. Returns (seconds per step,
        launch description, last loss, eager step callable)."""
        def fwd_bwd():
            loss = w.run(criterion=ops.cross_entropy_sum, mode=mode)["loss"]
            if w.model.compute_dtype == torch.float16:
                (loss * trainer.loss_scale).backward()           # the fused step divides the scale out again (and skips on overflow)
            else:
                loss.backward()
            return loss

        def eager():
            trainer.zero_grad()
            loss = fwd_bwd()
            trainer.allreduce_grads()
            trainer.step()
            return loss

        eager_s, loss = float("inf"), None
        for i in range(max(1, warmup)):
            tw = time.perf_counter()
            loss = eager()
            torch.cuda.synchronize()
            eager_s = min(eager_s, time.perf_counter() - tw)
            log(f"{what}: warmup {i}: {1e3 * (time.perf_counter() - tw):.1f} ms loss {float(loss.detach()):.5f}")
        step, launch = eager, "eager (one kernel launch per op from Python)"
        if world > 1:
            from vln_imagine_amd.train import sync_autotune
            sync_autotune(0)                            # every rank runs rank 0's kernel choices
        if graph:
            import gc
            loss = None                                 # drop the last eager autograd graph (its AccumulateGrad nodes) before capturing
            gc.collect()
            slow, captured, failed = True, None, None
            try:
                captured = trainer.capture(fwd_bwd, warmup=1)
            except Exception as e:                      # one GPU: keep measuring with the eager step and say so
                failed = f"{type(e).__name__}: {e}"
                log(f"{what}: graph capture failed ({failed})")
                if os.environ.get("VLNI_BENCH_TRACEBACK") == "1":
                    import traceback
                    traceback.print_exc()
            if world > 1 and agree(failed is not None):
                # several ranks: a silent fall-back would time a different program (44 instead of 33 ms) under the same headline - stop
                # instead (--no-graph asks for the eager step explicitly). Every rank reaches this decision BEFORE any replay runs (a
                # replay contains the gradient all-reduce: a rank that failed to capture must not be paired with it).
                raise SystemExit(f"bench.py: hipGraph capture failed on a rank of {world} ({failed}); rerun with --no-graph to time the eager step")
            if captured is not None:
                captured()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for _ in range(2):
                    captured()
                torch.cuda.synchronize()
                tg = (time.perf_counter() - tg) / 2
                log(f"{what}: captured into hipGraphs; replay {1e3 * tg:.1f} ms vs eager {1e3 * eager_s:.1f} ms")
                # never seen on a dedicated GPU; two processes SHARING one GPU replay pathologically slowly (VLNI_BENCH_KEEP_GRAPH=1: the
                # rehearsal of tests/bench_runner.py keeps the replay anyway - it checks the path, not the time)
                slow = tg > 1.3 * eager_s and os.environ.get("VLNI_BENCH_KEEP_GRAPH") != "1"
            if agree(slow):
                log(f"{what}: no usable graph replay on this box: timing the eager step")
            else:
                step = captured
                launch = "hipGraph replay (zero+fwd+bwd | wgrad flush%s | clip+AdamW)" % (
                    " in %d ranges, RCCL all-reduce of each range on a side stream under the next" % len(trainer.comm_ranges())
                    if (world > 1 or forced) else "")
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # one event per step boundary: the median step (8d)
        fence()
        t0 = time.perf_counter()
        marks[0].record()
        for i_ in range(steps):
            loss = step()
            marks[i_ + 1].record()
        fence()
        dt = dt_local = time.perf_counter() - t0
        per = sorted(marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(steps))
        median_ms[what] = round(per[len(per) // 2], 3)
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        log(f"{what}: timed {1e3 * dt / steps:.2f} ms/step")
        if (world > 1 or forced) and rccl is not None and "exchange" not in rccl:
            trainer.time_exchange = True                # one more, untimed step with HIP events around the exchange
            step()
            torch.cuda.synchronize()
            trainer.time_exchange = False
            rccl["exchange"] = trainer.exchange_report()
            # self-checks of the first real multi-GPU line: every rank's own step time, and what the collective costs the
            # step = timed step - the same pipeline (pack, side stream, unpack, graphs) with the all-reduce itself left out
            rccl["per_rank_ms_per_step"] = gather_floats(1e3 * dt_local / steps, world, dev) if world > 1 \
                else [round(1e3 * dt_local / steps, 3)]
            trainer.skip_exchange = True
            k2 = max(3, min(steps, 10))
            step(); fence()
            t1 = time.perf_counter()
            for _ in range(k2):
                step()
            fence()
            d2 = time.perf_counter() - t1
            trainer.skip_exchange = False
            if world > 1:
                t2 = torch.tensor([d2], device=dev, dtype=torch.float64)
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                d2 = float(t2.item())
            ex = rccl["exchange"] or {}
            rccl["ms_per_step_without_exchange"] = round(1e3 * d2 / k2, 3)
            rccl["allreduce_ms_exposed"] = round(1e3 * (dt / steps - d2 / k2), 3)
            rccl["payload_bytes_per_step"] = sum(ex.get("payload_bytes_per_range", [])) or None
            rccl["payload_dtype"] = ex.get("payload_dtype")
            step()                                      # gradients of every rank agree again before anything else runs
        return dt / steps, launch, float(loss.detach()), eager

    if args.load_tune:
        with open(args.load_tune) as fh:                  # JSON written by --dump-tune (ops.export_tune): plain data, no object deserialisation
            tn = json.load(fh)
        ops.import_tune(tn)
        log(f"kernel choices loaded from {args.load_tune}: {len(tn['gemm'])} GEMM shapes, {len(tn['tn']) + len(tn['tnb'])} weight-gradient classes")

    # ---- the metric's workload -------------------------------------------------------------------------------------------
    w = Workload(args.model, args, shipped, dev, dtype, tag=f"bench{rank}")
    if args.model == "hamt":
        w.model.visual_lang_rows = args.lang_rows
    if args.train_mode:
        w.model.train()
    trainer = FlatTrainer(w.model, lr=1e-5, grad_comm_dtype=comm, **scaler)
    log(f"model + episode ready on {dev}; warmup {args.warmup}, steps {args.steps}, dtype {args.dtype}, world {world}")
    sec, launch, last_loss, eager_step = measure(w, trainer, args.steps, args.warmup, mode=args.mode, graph=args.graph)
    ms = sec * 1e3
    eps = args.batch * world / sec
    if args.dump_tune and rank == 0:
        with open(args.dump_tune, "w") as fh:
            json.dump(ops.export_tune(), fh)

    # ---- roofline of the dominant kernel family (instrumented pass, not part of the timed region) -----------------------------
    def roofline_of(w, sec, eager_step, family, with_ceiling=True):
        """One instrumented (eager, HIP-event-timed) step of workload `w` -> the `roofline` object of its forward / dgrad GEMM family, plus the
        weight-gradient and attention families of the same step. Not part of any timed region."""
        ms = sec * 1e3
        from vln_imagine_amd import _lib
        rec, epi, fam, wg_bytes = [], [0.0], [], [0.0]
        orig_call = _lib.call
        FAM = ("vlni_gemm_tn_h16_grouped_part", "vlni_gemm_tn_h16_grouped_v", "vlni_reduce_parts_sq", "vlni_attn_bwd_dual", "vlni_attn_bwd", "vlni_attn_fwd_dual",
               "vlni_attn_fwd")

        def timed_call(name, *a_):              # the other kernel families of the step, by entry point (algorithmic work from the call's own arguments)
            if name not in FAM:
                return orig_call(name, *a_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r_ = orig_call(name, *a_)
            e1.record()
            es_ = 2
            if name.startswith("vlni_gemm_tn"):                     # (dtype, nseg, A, B, M[], lda/N, ldb/K, ...): 2 sum(M) N K flop
                n_, pm_ = a_[1], a_[4]
                rows_ = sum(pm_[i] for i in range(n_))
                work = 2.0 * rows_ * a_[9] * a_[10]
                wg_bytes[0] += 2.0 * rows_ * (a_[9] + a_[10]) + 4.0 * a_[9] * a_[10]        # dY and X read once (16-bit), dW written once (float32)
            elif name == "vlni_reduce_parts_sq":
                work = 0.0
            elif name.endswith("_dual"):                            # (..., B, nh, Sq[2], Sk[2], ...): bytes of q, k, v, out (+ dout, dq, dk, dv)
                off = 21 if "bwd" in name else 12
                B_, nh_, sq_, sk_ = a_[off], a_[off + 1], a_[off + 2], a_[off + 3]
                per = 4 if "bwd" in name else 2
                work = float(sum(B_ * (per * sq_[i] + per * sk_[i]) * nh_ * 64 * es_ for i in range(2)))
            else:
                off = 21 if "bwd" in name else 12
                B_, nh_, sq_, sk_ = a_[off], a_[off + 1], a_[off + 2], a_[off + 3]
                per = 4 if "bwd" in name else 2
                work = float(B_ * (per * sq_ + per * sk_) * nh_ * 64 * es_)
            fam.append((name, work, e0, e1))
            return r_

        orig, orig2 = ops.gemm_nt, ops.gemm_nt2
        orig, orig2 = ops.gemm_nt, ops.gemm_nt2

        def _epi_bytes(k):              # tensors the fused epilogue reads / writes besides C: residual, GELU' source, pre-activation
            n = 0
            for key in ("residual", "dact_src", "preact"):
                v = k.get(key)
                for t_ in (v if isinstance(v, (tuple, list)) else (v,)):
                    if torch.is_tensor(t_):
                        n += t_.numel() * t_.element_size()
            return n

        def _epi_tag(k):                # which epilogue kind the launch runs (the persistent kernels are instantiated per kind)
            return (" res" if k.get("residual") is not None else "") + (" drop" if k.get("drop") is not None else "") + (" pre" if k.get("preact") is not None else "")

        def _n_of(b):
            return b.n if isinstance(b, ops.WT) else b.t.shape[1] if isinstance(b, ops.KN) else b.shape[0]      # WT / KN: dgrad operand handles

        def timed(a, b, *p, **k):
            if ops._ghost():                           # the tape's ghost pass launches nothing: not a GEMM launch
                return orig(a, b, *p, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, *p, **k)
            e1.record()
            rec.append((2.0 * a.shape[0] * _n_of(b) * a.shape[1], e0, e1, (a.shape[0], _n_of(b), a.shape[1]),
                        ("nn" if isinstance(b, (ops.WT, ops.KN)) else "nt") + f" act{k.get('act', 0)} dact{k.get('dact', 0)}" + _epi_tag(k), _epi_bytes(k)))
            epi[0] += _epi_bytes(k)
            return r

        def timed2(a, b, *p, **k):                     # dual-problem launches (language + vision stream in one launch)
            if ops._ghost():
                return orig2(a, b, *p, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig2(a, b, *p, **k)
            e1.record()
            rows = a[0].shape[0] + a[1].shape[0]
            rec.append((2.0 * rows * _n_of(b[0]) * a[0].shape[1], e0, e1, (rows, _n_of(b[0]), a[0].shape[1]),
                        ("nn" if isinstance(b[0], (ops.WT, ops.KN)) else "nt") + f" act{k.get('act', 0)} dact{k.get('dact', 0)}" + _epi_tag(k)
                        + f" dual {a[0].shape[0]}+{a[1].shape[0]}", _epi_bytes(k)))
            epi[0] += _epi_bytes(k)
            return r

        ops.gemm_nt, ops.gemm_nt2, _lib.call = timed, timed2, timed_call
        # the block-level entry points issue a sublayer's launches from C, out of reach of the wrappers above: this one step takes the
        # launch-by-launch path (the same kernels with the same cached choices, one C call each)
        blk_was, ops.BLOCK_CALLS = ops.BLOCK_CALLS, False
        try:
            # keep the stream busy while the host enqueues the step, so that each event pair brackets the kernel alone and not
            # the host's gap between recording the event and launching (otherwise the average reads ~35 % above rocprof's)
            torch.cuda._sleep(int(0.05 * getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2.4e6) * 1e3))
            eager_step()
            torch.cuda.synchronize()
        finally:
            ops.gemm_nt, ops.gemm_nt2, _lib.call = orig, orig2, orig_call
            ops.BLOCK_CALLS = blk_was
        log("instrumented roofline step done")
        tot_f = sum(r[0] for r in rec)
        raw_ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        # What an event PAIR costs with nothing between its two records (4.6-4.8 us on this stack: the second marker's own turn in the queue) is in every
        # measurement above and not in the kernel: measured here the same way (busy stream, host ahead), taken off each launch. With it the average agrees
        # with rocprofv3's kernel durations (profiles/r03_step_breakdown.md) to 1-2 %; without it the events read ~10 % long on 50-us launches.
        cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        torch.cuda._sleep(int(0.02 * getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2.4e6) * 1e3))
        for c0_, c1_ in cal:
            c0_.record()
            c1_.record()
        torch.cuda.synchronize()
        pair_ms = sorted(c0_.elapsed_time(c1_) for c0_, c1_ in cal)[len(cal) // 2]
        tot_ms = sum(max(r[1].elapsed_time(r[2]) - pair_ms, 1e-4) for r in rec)
        if os.environ.get("VLNI_GEMM_BREAKDOWN"):
            agg = {}
            for f, e0, e1, shp, kind, _ in rec:
                a_ = agg.setdefault((shp, kind), [0, 0.0, 0.0])
                a_[0] += 1; a_[1] += max(e0.elapsed_time(e1) - pair_ms, 1e-4); a_[2] += f
            for (shp, kind), (n, ms_, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ["VLNI_GEMM_BREAKDOWN"])]:
                log(f"gemm M={shp[0]:6d} N={shp[1]:5d} K={shp[2]:5d} {kind:34s}: {n:4d} calls {ms_:7.3f} ms {ms_ / n * 1e3:7.1f} us each {f / ms_ / 1e9:7.1f} TF/s"
                    f"  over 0.32 of peak: {ms_ - f / (0.32 * peak * 1e9):6.3f} ms")
        ach = tot_f / (tot_ms * 1e-3) / 1e12
        traffic, traffic_src, wg_traffic = None, None, None
        try:      # HBM bytes per launch: separate rocprofv3 --pmc passes of this command (never collected inside a timed run)
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                         if f.endswith("_pmc_traffic.json") and ("duet" in f) == (family == "duet"))
            if pmc and args.dtype == "bf16" and args.mode == "taped" and args.train_mode:      # collected on the default workloads only
                pj = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
                traffic = round(pj["bytes_per_launch"])
                tn_ = [v for k_, v in pj.get("per_family_kb_per_launch", {}).items() if k_.startswith("gemm_tn")]
                if tn_:                            # weight-gradient kernels: FETCH_SIZE doubled (gfx950 correction) + WRITE_SIZE, per launch
                    wg_traffic = round(sum((2.0 * v["fetch_raw"] + v["write"]) * 1024.0 * v["launches"] for v in tn_) / sum(v["launches"] for v in tn_))
                traffic_src = f"profiles/{pmc[-1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)"
        except Exception:
            traffic = None
        # what a large dense GEMM reaches on THIS box ("a measured large-GEMM ceiling"): the vendor library on 8192^3 in the
        # timed dtype, random operands - a reference measurement only, the product path never calls it
        ceiling, ceiling_at = None, None
        try:
            if not with_ceiling:
                raise StopIteration
            cd = torch.float32 if args.dtype == "fp32" else dtype
            for n_ in ((2048, 4096) if args.dtype == "fp32" else (4096, 8192)):
                xa, xb = torch.randn(n_, n_, device=dev).to(cd), torch.randn(n_, n_, device=dev).to(cd)
                for _ in range(2):
                    torch.matmul(xa, xb.t())
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(5):
                    torch.matmul(xa, xb.t())
                c1.record()
                torch.cuda.synchronize()
                tf = round(5 * 2.0 * n_ ** 3 / (c0.elapsed_time(c1) * 1e-3) / 1e12, 1)
                if ceiling is None or tf > ceiling:
                    ceiling, ceiling_at = tf, n_
                del xa, xb
        except StopIteration:
            pass
        except Exception as e:                                  # never lets the reference measurement break the line
            log(f"large-GEMM ceiling not measured ({type(e).__name__}: {e})")
        alg = sum(2.0 * (m_ * k_ + n_ * k_ + m_ * n_) for _, _, _, (m_, n_, k_), _, _ in rec)
        # Per shape: the launch's floor is max(MFMA time, HBM time) - M x 768 x 768 with a residual or GELU' operand is HBM-bound, not MFMA-bound
        # (VERDICT round 5, item 5). flops / peak against bytes (A + B + C + the epilogue's tensors, element size of the timed dtype) / the HBM
        # peak of the guide (8 TB/s; 6.29 TB/s is what a copy reaches: `hbm_achievable_us`).
        es_b = 4 if args.dtype == "fp32" else 2
        shp_agg = {}
        for f, e0, e1, (m_, n_, k_), kind, eb in rec:
            a_ = shp_agg.setdefault((m_, n_, k_, kind), [0, 0.0, 0.0, 0.0])
            a_[0] += 1; a_[1] += max(e0.elapsed_time(e1) - pair_ms, 1e-4); a_[2] += f; a_[3] += es_b * (m_ * k_ + n_ * k_ + m_ * n_) + eb
        per_shape, floor_ms, floor_ach_ms = [], 0.0, 0.0
        for (m_, n_, k_, kind), (cnt, ms_, f, by) in sorted(shp_agg.items(), key=lambda kv: -kv[1][1]):
            mfma_us, hbm_us, hbm_ach_us = f / cnt / (peak * 1e12) * 1e6, by / cnt / (HBM_PEAK_GBS * 1e9) * 1e6, by / cnt / 6.29e12 * 1e6
            floor_ms += cnt * max(mfma_us, hbm_us) * 1e-3
            floor_ach_ms += cnt * max(mfma_us, hbm_ach_us) * 1e-3
            per_shape.append({"M": m_, "N": n_, "K": k_, "kind": kind, "launches": cnt, "us": round(ms_ / cnt * 1e3, 1), "tflops": round(f / ms_ / 1e9, 1),
                              "mfma_us": round(mfma_us, 1), "hbm_us": round(hbm_us, 1), "bound": "hbm" if hbm_us > mfma_us else "mfma",
                              "frac_of_floor": round(max(mfma_us, hbm_us) / (ms_ / cnt * 1e3), 3)})
        if os.environ.get("VLNI_GEMM_SHAPES_OUT") and with_ceiling:     # (the line's own workload, not the DUET extra) tools/gemm_shapes.py replays these shapes beside the vendor library
            json.dump(per_shape, open(os.environ["VLNI_GEMM_SHAPES_OUT"], "w"), indent=1)
        def fam_sum(prefixes):
            sel = [f_ for f_ in fam if f_[0].startswith(prefixes)]
            t_ = sum(max(f_[2].elapsed_time(f_[3]) - pair_ms, 1e-4) for f_ in sel)
            return len(sel), sum(f_[1] for f_ in sel), t_
        families = {}
        n_, f_, t_ = fam_sum(("vlni_gemm_tn",))
        nr_, _, tr_ = fam_sum(("vlni_reduce_parts_sq",))
        if n_:
            families["weight_gradients"] = {"bound": "mfma", "launches": n_, "ms": round(t_, 3), "achieved": round(f_ / (t_ * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                                            "frac": round(f_ / (t_ * 1e-3) / 1e12 / peak, 4), "partial_reduction_ms": round(tr_, 3),
                                            # dY and X read once, dW written once per launch; traffic from the same --pmc passes as `roofline.traffic`
                                            "algorithmic_bytes": round(wg_bytes[0] / n_), "traffic": wg_traffic,
                                            "traffic_ratio": round(wg_traffic / (wg_bytes[0] / n_), 3) if wg_traffic else None}
        for key_, pre_ in (("attention_bwd", ("vlni_attn_bwd",)), ("attention_fwd", ("vlni_attn_fwd",))):
            n_, by_, t_ = fam_sum(pre_)
            if n_:
                families[key_] = {"bound": "hbm", "launches": n_, "ms": round(t_, 3), "achieved": round(by_ / (t_ * 1e-3) / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                  "frac": round(by_ / (t_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "bytes": "q, k, v, out read/written once" + (" + dout read, dq, dk, dv written" if "bwd" in key_ else "")}
        roof = {"bound": "mfma", "kernel": "gemm_nt_* / gemm_nn_* <%s>" % {"bf16": "__bf16", "fp16": "_Float16", "fp32": "float"}[args.dtype],
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                # the family against the per-shape floor max(flops / MFMA peak, bytes / HBM peak): 1.0 = every launch at its own roofline
                "frac_of_per_shape_floor": round(floor_ms / tot_ms, 4), "frac_of_per_shape_floor_hbm_6p29": round(floor_ach_ms / tot_ms, 4),
                "hbm_bound_launches": sum(x["launches"] for x in per_shape if x["bound"] == "hbm"),
                "bound_per_shape": per_shape[:12], "traffic": traffic,
                "traffic_source": traffic_src, "traffic_ratio": round(traffic / (alg / len(rec)), 3) if traffic else None,
                # the same against A + B + C + what the fused epilogues read / write besides C; the rest is the weight panel fetched once per XCD's
                # L2 (8 copies from the Infinity Cache: profiles/r06_gemm_traffic.md)
                "traffic_ratio_with_epilogue_operands": round(traffic / ((alg + epi[0]) / len(rec)), 3) if traffic else None,
                "algorithmic_bytes_per_launch": round(alg / len(rec)),
                "algorithmic_bytes_per_launch_with_epilogue": round((alg + epi[0]) / len(rec)),
                "launches_per_step": len(rec), "avg_launch_us": round(tot_ms * 1e3 / len(rec), 2),
                "event_pair_overhead_us": round(pair_ms * 1e3, 2),
                "achieved_uncorrected": round(tot_f / (raw_ms * 1e-3) / 1e12, 2), "avg_launch_us_uncorrected": round(raw_ms * 1e3 / len(rec), 2),
                "avg_gflop_per_launch": round(tot_f / len(rec) / 1e9, 3),
                "gemm_share_of_step": round(tot_ms / ms, 3),
                "step_algorithmic_tflops": round(w.flops / sec / 1e12, 2),
                "step_frac_of_peak": round(w.flops / sec / 1e12 / peak, 4),
                "measured_large_gemm_tflops": ceiling,
                "frac_of_measured_large_gemm": round(ach / ceiling, 4) if ceiling else None,
                "measured_large_gemm": f"torch.matmul(a, b.T) (vendor library), {ceiling_at}^3 {args.dtype}, random operands, best of 4096^3 / 8192^3 "
                                       f"(fp32: 2048^3 / 4096^3), this box" if with_ceiling else None,
                "families": families or None}
        return roof

    roof = None
    if not args.no_roofline:           # EVERY rank runs the instrumented step (it contains the gradient all-reduce)
        roof = roofline_of(w, sec, eager_step, args.model)

    # ---- error of the timed path against the fp32 parity path (same model, same weights, B = 8 slice, fwd + bwd) -------------
    parity = None
    if rank == 0 and not args.no_parity and args.dtype in ("bf16", "fp16"):
        ws = Workload(args.model, args, shipped, dev, dtype, batch=8, tag="slice", model=w.model)
        w.model.eval()                                            # p = 0 on both sides: the distance of the arithmetic, not of two mask draws
        w32 = ws.build(dev, torch.float32)
        w32.load_state_dict(w.model.state_dict())                 # the timed model has taken optimizer steps: compare at ITS weights
        trainer.zero_grad()
        o16 = ws.run(criterion=ops.cross_entropy_sum, keep=True)
        if dtype == torch.float16:                                # backward on S * loss as in a training step, then divide S out
            (o16["loss"] * trainer.loss_scale).backward()
            trainer.flush()
            trainer.flat_g.mul_(1.0 / float(trainer.state[4]))
        else:
            o16["loss"].backward()
            trainer.flush()
        o32 = ws.run(criterion=ops.cross_entropy_sum, keep=True, model=w32)
        o32["loss"].backward()
        parity = compare_runs(o16, o32, dict(w.model.named_parameters()), dict(w32.named_parameters()), ws.logits_key)
        parity = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in parity.items()}
        parity["sample"] = (f"B=8, T={args.T}, fwd+bwd at the timed model's current weights; fp32 = the exact-fp32 MFMA path held to the "
                            "reference goldens at 1e-4")
        # what the 16-bit all-reduce payload of N > 1 adds on top: the float32 gradient arena of this slice through the pack /
        # unpack kernels of the exchange (vlni_scale_cast, 1/N pre-division folded in) - per rank; a ring sum of N such payloads adds at most sqrt(N) of it
        g32 = trainer.flat_g
        nrm = float(g32.norm())
        if nrm > 0:
            pk = torch.empty(g32.numel(), dtype=torch.bfloat16, device=g32.device)
            back = torch.empty_like(g32)
            st_ = torch.cuda.current_stream().cuda_stream
            from vln_imagine_amd import _lib
            _lib.call("vlni_scale_cast", 0, 1, g32.data_ptr(), pk.data_ptr(), g32.numel(), 0.125, st_)
            _lib.call("vlni_scale_cast", 1, 0, pk.data_ptr(), back.data_ptr(), g32.numel(), 8.0, st_)
            parity["grad_allreduce_bf16_payload_rel_l2"] = round(float((back - g32).norm()) / nrm, 6)
            del pk, back
        log(f"bf16 vs fp32: {parity}")
        del w32, o16, o32, ws
        if args.train_mode:
            w.model.train()
    if world > 1 or forced:
        dist.barrier()

    # ---- extra lines, reported beside `value` (never instead of it) -------------------------------------------------------
    extras = {}
    if world == 1 and not forced and not args.no_extras:
        k_extra = max(3, min(args.steps, 20))

        def line(sec_, flops_, note):
            return {"value": round(args.batch / sec_, 2), "unit": "episodes/s", "ms_per_step": round(sec_ * 1e3, 3),
                    "step_algorithmic_tflops": round(flops_ / sec_ / 1e12, 2), "note": note}

        if args.model == "duet" and args.mode != "stepwise":
            s_, _, _, _ = measure(w, trainer, k_extra, 2, mode="stepwise", graph=args.graph, what="stepwise")
            extras["stepwise"] = line(s_, w.flops, "one autograd graph per `panorama` / `navigation` call (rounds 1-2's path)")
        if args.model == "duet" and args.mode != "time_batched":
            s_, _, _, _ = measure(w, trainer, k_extra, 2, mode="time_batched", graph=args.graph, what="time_batched")
            extras["time_batched"] = line(s_, w.flops, "forward AND backward on T x B samples, maps padded to the episode's largest (teacher forcing only); same results")
        if args.model == "hamt":
            for md, note in (("stepwise", "one autograd graph per `visual` / `history` call (rounds 1-2's headline path): T x shorter backward launches"),
                             ("time_batched", "forward AND backward on T x B samples (teacher forcing only); same results"),
                             ("taped", "step-by-step forward calls on an episode tape + ONE episode-batched backward")):
                if md != args.mode:
                    s_, _, _, _ = measure(w, trainer, k_extra, 2, mode=md, graph=args.graph, what=md)
                    extras[md] = line(s_, w.flops, note)
        if args.model == "hamt" and args.graph:
            # a SAMPLED rollout's launch pattern: T + 2 graph replays with the host in between (it reads step t's logits - one sync per
            # step - and only then provides step t + 1's observation and step t's history features), ONE episode-batched backward
            from vln_imagine_amd.hamt.buckets import EpisodeBuffers, SteppedEpisodeGraphs
            import gc
            gc.collect()
            bufs = EpisodeBuffers(args.batch, args.L, args.V, args.I, args.T, dev).load(w.et.ep)
            sg = SteppedEpisodeGraphs(trainer, w.model, bufs)

            def sampled_episode():
                sg.begin()
                acts = []
                for t in range(args.T):
                    for k in EpisodeBuffers.OBS_KEYS:
                        bufs.steps[t][k].copy_(w.et.steps[t][k])
                    if t > 0:
                        for k in EpisodeBuffers.HIST_KEYS:
                            bufs.steps[t - 1][k].copy_(w.et.steps[t - 1][k])
                    sg.step(t)
                    acts.append(sg.logits(t).argmax(1).cpu())             # the host decides where to go
                for k in EpisodeBuffers.HIST_KEYS:
                    bufs.steps[args.T - 1][k].copy_(w.et.steps[args.T - 1][k])
                return sg.finish()

            sampled_episode()
            fence()
            t0 = time.perf_counter()
            for _ in range(k_extra):
                sampled_episode()
            fence()
            s_ = (time.perf_counter() - t0) / k_extra
            log(f"sampled-stepped: timed {1e3 * s_:.2f} ms/step")
            extras["sampled_stepped"] = line(s_, w.flops, "a sampled rollout's pattern (hamt.buckets.SteppedEpisodeGraphs): begin | T step graphs | ghost + "
                                                          "backward + optimizer, the host reading each step's logits (argmax, one sync) before it writes the next "
                                                          "observation; history call lags one step (no side-stream overlap); same episode-batched backward")
            del sg, bufs
            gc.collect()
        if args.model == "duet" and args.graph:
            # DUET's sampled-rollout pattern: T + 2 graph replays; before step t the host reads step t - 1's logits (one sync) and writes step t's
            # panorama, map tensors, node sources and fusion plan into the static buffers (host-built numpy -> device copies, as the agent would)
            from vln_imagine_amd.duet.buckets import DuetEpisodeBuffers, SteppedEpisodeGraphs as DuetStepped
            import gc
            gc.collect()
            gmax = max(s_["gmap_masks"].shape[1] for s_ in w.et.ep.steps)
            dbufs = DuetEpisodeBuffers(args.batch, args.L, args.I, args.T, gmax, dev).load(w.et.ep)
            dsg = DuetStepped(trainer, w.model, dbufs)

            def duet_sampled_episode():
                dsg.begin()
                for t in range(args.T):
                    dbufs.put_step(t, w.et.ep.steps[t])
                    dsg.step(t)
                    dsg.logits(t).argmax(1).cpu()
                return dsg.finish()

            duet_sampled_episode()
            fence()
            t0 = time.perf_counter()
            for _ in range(k_extra):
                duet_sampled_episode()
            fence()
            s_ = (time.perf_counter() - t0) / k_extra
            log(f"sampled-stepped: timed {1e3 * s_:.2f} ms/step")
            extras["sampled_stepped"] = line(s_, w.flops, "a sampled rollout's pattern (duet.buckets.SteppedEpisodeGraphs): begin | T step graphs | ghost + backward + "
                                                          "optimizer; before every step the host reads the previous logits (one sync), builds the step's map tensors / node "
                                                          "sources / fusion plan in numpy and copies them with the panorama features (3.5 MB) into the static buffers")
            del dsg, dbufs
            gc.collect()
        if args.model == "hamt" and args.lang_rows == "all":
            w.model.visual_lang_rows = "cls"
            s_, _, _, _ = measure(w, trainer, k_extra, 2, mode=args.mode, graph=args.graph, what="cls-rows")
            w.model.visual_lang_rows = "all"
            extras["cls_rows"] = line(s_, w.flops, "what the VLNBertCMT wrapper runs: the last cross-modal layer computes only the language [CLS] row it "
                                                   "reads (NavCMT.visual_lang_rows = 'cls'); logits, loss and gradients identical "
                                                   "(tests/test_hamt_gpu.py), step_algorithmic_tflops still counts the reference's full rows")
        if args.train_mode:
            w.model.eval()
            s_, _, _, _ = measure(w, trainer, k_extra, 2, mode=args.mode, graph=args.graph, what="eval-mode")
            w.model.train()
            extras["eval_mode"] = line(s_, w.flops, "model.eval(): dropout p = 0 (what rounds 1-2 quoted as the headline)")
        else:
            w.model.train()
            s_, _, _, _ = measure(w, trainer, k_extra, 2, mode=args.mode, graph=args.graph, what="train-mode")
            w.model.eval()
            extras["train_mode"] = line(s_, w.flops, "model.train(): attention-probability and hidden dropout p = 0.1 inside the fused kernels "
                                                     "(masks regenerated in backward), the reference's training mode")
        if args.model == "hamt":
            w1 = Workload("hamt", args, shipped, dev, dtype, T=1, tag="benchT1", model=w.model)
            s_, _, _, _ = measure(w1, trainer, k_extra, 2, mode=args.mode, graph=args.graph, what="T=1")
            extras["T1"] = line(s_, w1.flops, "one navigation step per episode (SURVEY 8d's second episode length)")
        if args.dtype == "bf16":
            w.model.set_compute_dtype(torch.float32)
            s_, _, _, _ = measure(w, trainer, 3, 1, mode=args.mode, graph=False, what="fp32")
            w.model.set_compute_dtype(dtype)
            e_ = line(s_, w.flops, "the parity path: every contraction on v_mfma_f32_32x32x2_f32 (exact fp32), eager launches")
            e_["step_frac_of_fp32_mfma_peak"] = round(w.flops / s_ / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
            extras["fp32"] = e_
        if args.dtype == "bf16":
            # BASELINE.json configs[4] names fp16 (the reference's only mixed precision: torch.cuda.amp in VLN-DUET/pretrain_src/
            # train_r2r.py:201-234): the same step on the f16 MFMA with dynamic loss scaling
            trainer.close()
            w16 = Workload(args.model, args, shipped, dev, torch.float16, tag=f"bench{rank}")
            tr16 = FlatTrainer(w16.model, lr=1e-5, grad_comm_dtype=comm, loss_scale=16384.0, growth_interval=2000)
            trainer = tr16                                   # measure()'s closures read `trainer`
            if args.train_mode:
                w16.model.train()
            s_, _, _, _ = measure(w16, tr16, k_extra, 2, mode=args.mode, graph=args.graph, what="fp16")
            e_ = line(s_, w16.flops, "float16 compute + GradScaler-style dynamic loss scaling in the fused step (float32 masters, f16 mirror)")
            e_["skipped_steps"] = int(float(tr16.state[5]))
            e_["loss_scale"] = float(tr16.state[4])
            extras["fp16"] = e_
            tr16.close()
            del w16
        if not shipped:
            trainer.close()
            del trainer
            ws_ = Workload(args.model, args, True, dev, dtype, tag=f"bench{rank}")
            tr2 = FlatTrainer(ws_.model, lr=1e-5, grad_comm_dtype=comm)
            if args.train_mode:
                ws_.model.train()
            s_, _, _, _ = measure(ws_, tr2, k_extra, 2, mode=args.mode, graph=args.graph, what="freeze=shipped")
            extras["freeze_shipped"] = line(s_, ws_.flops, "the released run's freeze: language stack (and HAMT history encoder) forward only")
            tr2.close()

    # BASELINE.json configs[3] (DUET, batch 32, one GPU) beside the metric's HAMT line, so that the driver's own `bench.py --gpus 1` times it:
    # the same taped step (step-by-step forward, one episode-batched backward, hipGraph replay), its GEMM family against the same peak
    if world == 1 and not forced and not args.no_extras and args.model == "hamt" and args.dtype == "bf16" and not shipped:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        wd = Workload("duet", args, False, dev, dtype, batch=32, tag=f"bench{rank}")
        if args.train_mode:
            wd.model.train()
        trd = FlatTrainer(wd.model, lr=1e-5, grad_comm_dtype=comm)
        trainer = trd                                        # measure()'s closures read `trainer`
        kd = max(3, min(args.steps, 20))
        s_, launch_d, _, eager_d = measure(wd, trd, kd, 3, mode="taped", graph=args.graph, what="duet_b32")
        e_ = {"value": round(32 / s_, 2), "unit": "episodes/s", "ms_per_step": round(s_ * 1e3, 3), "ms_per_step_median": median_ms.get("duet_b32"),
              "steps": kd, "step_algorithmic_tflops": round(wd.flops / s_ / 1e12, 2), "launch": launch_d,
              "workload": f"{wd.label}, batch 32, {args.L} text, 36 views, {args.I} imaginations, T={args.T}, freeze=none, "
                          + ("train mode: in-kernel dropout p=0.1" if args.train_mode else "eval") + ", episode tape",
              "note": "BASELINE.json configs[3]; `python bench.py --model duet` is the full line (roofline traffic, CPU baseline, extras)"}
        if not args.no_roofline:
            rd = roofline_of(wd, s_, eager_d, "duet", with_ceiling=False)
            e_["roofline"] = {k_: rd[k_] for k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "launches_per_step",
                                                     "avg_launch_us", "gemm_share_of_step", "step_frac_of_peak", "families")}
        extras["duet_b32"] = e_
        trd.close()
        del wd, trd

    # The number an UNCHANGED reference agent gets: the VLNBertCMT / VLNBert wrappers called eagerly mode by mode,
    # plain autograd, clip_grad_norm_, torch.optim.AdamW - no episode tape, no FlatTrainer, no captured graph (vln_imagine_amd/dropin.py =
    # agent_cmt.py:400-700,827-832 / agent.py:409-500 + agent_base.py:223-228). Host-bound: wall clock over whole iterations.
    if world == 1 and not forced and not args.no_extras and not shipped:
        import gc
        from vln_imagine_amd import dropin, graphed
        drop = {}
        for fam, bsz in ((("hamt", args.batch), ("duet", 32)) if (args.model == "hamt" and args.dtype == "bf16") else ((args.model, args.batch),)):
            for per_call_graphs in (True, False):
                # True = what the wrappers do by default since round 6 (vln_imagine_amd/graphed.py: one autograd node per wrapper call, replaying
                # captured forward / backward graphs); False = the same agent code with VLNI_GRAPHED_MODES=0 (rounds 4-5: every kernel a Python ->
                # C-ABI crossing, host-bound)
                gc.collect()
                torch.cuda.empty_cache()
                was_g, graphed.ENABLED = graphed.ENABLED, per_call_graphs and graphed.ENABLED
                try:
                    wx = Workload(fam, args, False, dev, dtype, batch=bsz, tag=f"bench{rank}")
                    if args.train_mode:
                        wx.model.train()
                    tr = dropin.DropInTrainer((dropin.wrap_hamt if fam == "hamt" else dropin.wrap_duet)(wx.model, feat_dropout=0.4 if args.train_mode else 0.0),
                                              wx.et, fam)
                    for _ in range(4):
                        tr.step()
                    torch.cuda.synchronize()
                    kx = 8
                    t0 = time.perf_counter()
                    for _ in range(kx):
                        loss_x = tr.step()
                    torch.cuda.synchronize()
                    sx = (time.perf_counter() - t0) / kx
                    st_g = dict(graphed.of(wx.model).stats)
                finally:
                    graphed.ENABLED = was_g
                    ops.set_seed_base(None)
                log(f"drop-in ({fam}, batch {bsz}, per-call graphs {'on' if per_call_graphs else 'off'}): {1e3 * sx:.2f} ms per iteration, "
                    f"loss {float(loss_x):.4f}, {st_g}")
                e_ = {"value": round(bsz / sx, 2), "unit": "episodes/s", "ms_per_step": round(1e3 * sx, 3), "batch": bsz, "iterations": kx,
                      "step_algorithmic_tflops": round(wx.flops / sx / 1e12, 2)}
                if per_call_graphs:
                    drop[fam] = dict(e_, wrapper_calls=st_g)
                else:
                    drop[fam]["without_per_call_graphs"] = e_
                del tr, wx
        drop["note"] = ("an unchanged reference agent on the drop-in modules (Seq2SeqCMTAgent / GMapNavAgent call pattern, vln_imagine_amd/dropin.py): wrappers "
                        "called per mode / step, loss.backward(), clip_grad_norm_, torch.optim.AdamW. Round 6: each wrapper call is one autograd node that "
                        "replays captured forward / backward graphs from its second sighting on (fixed shapes; ragged batches stay eager); "
                        "`without_per_call_graphs` = VLNI_GRAPHED_MODES=0, the host-bound path of rounds 4-5. The headline needs the trainer of "
                        "INTEGRATION.md section 4")
        extras["drop_in_eager"] = drop

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, shipped)

    if rank == 0:
        line_ = {
            "metric": "episodes/sec (fwd+bwd) HAMT-Imagine 9L, batch 64" if args.model == "hamt"
            else "episodes/sec (fwd+bwd) DUET-Imagine 9L+2pano+4+4X, batch 32", "value": round(eps, 2), "unit": "episodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_median": median_ms.get("step"),            # HIP events at the step boundaries of the same timed region (this rank)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{w.label}, batch {args.batch}/GPU, {args.L} text, {args.V} obs tokens, "
                                   f"{args.I} imaginations, T={args.T} steps/episode, freeze={args.freeze}, "
                                   + ("train mode: in-kernel dropout p=0.1" if args.train_mode else "dropout p=0 (eval)")
                                   + {"time_batched": ", steps time-batched (teacher forcing)", "stepwise": ", step-by-step calls (one autograd graph each)",
                                      "taped": ", step-by-step forward calls, ONE episode-batched backward (episode tape)"}[args.mode]
                                   + (", last X-layer: language [CLS] row only" if args.model == "hamt" and args.lang_rows == "cls" else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "launch": launch, "rccl": rccl,
                       "grad_allreduce": (args.grad_comm + " payload, flat arena, flush -> all-reduce pipeline, RCCL") if (world > 1 or forced) else "none (1 GPU)",
                       "steps_per_sec": round(args.T * args.batch * world / sec, 1),
                       "loss": round(last_loss, 5)},
            "argv": " ".join(sys.argv[1:]),
            "roofline": roof, "cpu_baseline": cpu, "bf16_vs_fp32": parity, "extras": extras or None,
            "time_batched": extras.get("time_batched"),
        }
        print(json.dumps(line_), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
