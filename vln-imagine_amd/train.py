"""Optimizer side of the measured step, MI355X-first: all trainable parameters and their gradients
live in two flat float32 arenas (one memset to zero the grads, one fused clip+AdamW kernel over
everything, gradient all-reduce over RCCL in a few large chunks instead of per-tensor buckets).

Restates the tail of Seq2SeqCMTAgent.train (VLN-HAMT/finetune_src/r2r/agent_cmt.py:809-832):
zero_grad, backward, clip_grad_norm_(40.), AdamW step; and DDP's gradient averaging (:61-63)."""
import torch
import torch.distributed as dist

from . import _lib, ops


def allreduce_mean_(flat, chunk_elems, comm_dtype=None):
    """In-place mean over the data-parallel group (DDP's gradient averaging, r2r/agent_cmt.py:61-63) on ONE flat
    buffer: a handful of large all-reduces (RCCL over xGMI on the GPU box, gloo in the CPU tests) instead of
    per-tensor buckets. comm_dtype=torch.bfloat16 halves the bytes on the xGMI links (the reduction then runs in bf16,
    like DDP's bf16 compression hook; the arena stays float32). No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    ws = dist.get_world_size()
    if comm_dtype is None or comm_dtype == flat.dtype:
        works = [dist.all_reduce(flat[o:o + chunk_elems], op=dist.ReduceOp.SUM, async_op=True)
                 for o in range(0, flat.numel(), chunk_elems)]
        for w in works:
            w.wait()
        return flat.mul_(1.0 / ws)
    # pre-divide so the bf16 sum of ws terms stays in range, reduce compressed chunks, expand back
    parts = []
    for o in range(0, flat.numel(), chunk_elems):
        c = (flat[o:o + chunk_elems] * (1.0 / ws)).to(comm_dtype)
        parts.append((o, c, dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True)))
    for o, c, w in parts:
        w.wait()
        flat[o:o + c.numel()].copy_(c)
    return flat


class FlatTrainer:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=40.0,
                 chunk_mb=128, grad_comm_dtype=None):
        self.model = model
        # arena order: the q/k/v projections of every attention module sit back to back (weights, then biases), so
        # the packed [2304,768] QKV gradient is ONE wgrad GEMM into a contiguous view (ops._packed_grad)
        order, seen = [], set()
        for mod in model.modules():
            if all(hasattr(mod, n) for n in ("query", "key", "value")):
                for attr in ("weight", "bias"):
                    for n in ("query", "key", "value"):
                        p = getattr(getattr(mod, n), attr)
                        if p is not None and p.requires_grad and id(p) not in seen:
                            order.append(p); seen.add(id(p))
        for p in model.parameters():
            if p.requires_grad and id(p) not in seen:
                order.append(p); seen.add(id(p))
        self.params = order
        dev = self.params[0].device
        assert dev.type == "cuda", "FlatTrainer needs the model on the GPU"
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8              # slots aligned to 16 bytes in the bf16 mirror too (32 B in float32)
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.hp = (lr, betas[0], betas[1], eps, weight_decay)
        self.max_norm = max_norm
        self.step_no = 0
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # everything that changes from step to step lives on the device, so a captured step replays correctly:
        # state = [clip factor, 1-beta1^t, 1-beta2^t, t]; lr_dev[0] = learning rate (set_lr() for schedules)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.chunk = chunk_mb * (1 << 20) // 4
        self.grad_comm_dtype = grad_comm_dtype
        # bf16 mirror of the parameter arena, kept current by the AdamW kernel (ops.ShadowCache hands out views of it)
        self.flat_b = torch.empty(n, dtype=torch.bfloat16, device=dev)
        ops.SHADOWS.set_arena(self.flat_p, self.flat_b)
        ops.DIRECT_GRAD = True
        ops.DEFER_WGRAD = True

    def zero_grad(self):
        ops._WQ.clear()
        self.flat_g.zero_()

    def flush(self):
        """Completes the gradient arena (deferred grouped weight-gradient GEMMs)."""
        ops.flush_wgrads()

    def allreduce_grads(self):
        self.flush()
        allreduce_mean_(self.flat_g, self.chunk, self.grad_comm_dtype)

    def set_lr(self, lr):
        self.lr_dev.fill_(lr)

    def close(self):
        """Undoes the process-wide switches the constructor flipped (direct gradient accumulation, deferred weight gradients,
        the registered parameter arena, a registered dropout seed base). The parameters keep pointing into the arenas."""
        ops._WQ.clear()
        ops._PART_BUFS.clear()                 # row-split workspaces and reduction tables of this arena's gradients
        ops._PART_TABLES.clear()
        ops.DIRECT_GRAD = ops.DEFER_WGRAD = False
        if ops.SHADOWS.arena is not None and ops.SHADOWS.arena[0] is self.flat_p:
            ops.SHADOWS.set_arena(None, None)
        ops.set_seed_base(None)

    def step(self):
        self.flush()
        st = torch.cuda.current_stream().cuda_stream
        self.step_no += 1
        _, b1, b2, eps, wd = self.hp
        self.sumsq.zero_()
        _lib.call("vlni_sumsq", self.flat_g.data_ptr(), self.n, self.sumsq.data_ptr(), st)
        _lib.call("vlni_optim_prepare", self.sumsq.data_ptr(), self.max_norm, b1, b2, self.state.data_ptr(), st)
        _lib.call("vlni_adamw_step_dev", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                  self.flat_b.data_ptr(), self.n, self.lr_dev.data_ptr(), b1, b2, eps, wd, self.state.data_ptr(), st)
        ops.SHADOWS.invalidate(optimizer_step=ops.SHADOWS.arena is not None and ops.SHADOWS.arena[0] is self.flat_p)

    def grad_norm(self):
        return float(self.sumsq.sqrt())

    def capture(self, fwd_bwd, warmup=1):
        """Captures the training step into two hipGraphs and returns a callable that replays them.

        fwd_bwd() runs forward + backward (on fixed-shape, fixed-address inputs; refill them in place between calls) and
        returns the loss tensor. Graph 1 = zero_grad + fwd_bwd + deferred weight-gradient flush (including the bf16 shadow
        refresh of every parameter), graph 2 = clip + AdamW; the gradient all-reduce runs eagerly between the two, so RCCL is
        never part of a capture. `warmup` REAL steps run first on a side stream (lazy initialisation, GEMM autotune).
        Drop every reference to earlier losses / outputs before calling this: an autograd graph that is still alive keeps its
        AccumulateGrad nodes on the old stream, which breaks the capture.
        One replay costs two graph launches instead of ~3000 kernel launches from Python."""
        return GraphedStep(self, fwd_bwd, warmup)


class GraphedStep:
    def __init__(self, trainer, fwd_bwd, warmup=1):
        import gc
        self.trainer = trainer
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                trainer.zero_grad()
                fwd_bwd()
                trainer.allreduce_grads()
                trainer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.SHADOWS.prepare()                            # device table of the batched transposed-shadow refresh
        ops.SHADOWS.invalidate(optimizer_step=True)      # the transposed-shadow rebuild must be recorded inside graph 1
        # dropout: the seeds recorded in the graph are constants, their device-resident base moves on every replay
        self.seed_base = torch.zeros(1, dtype=torch.int32, device=trainer.flat_p.device)
        ops.set_seed_base(self.seed_base)
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb):
            self.seed_base.add_(7919)
            trainer.zero_grad()
            self.loss = fwd_bwd()
            trainer.flush()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool()):
            trainer.step()
        trainer.step_no -= 1                   # recorded, not executed

    def __call__(self):
        self.g_fb.replay()
        allreduce_mean_(self.trainer.flat_g, self.trainer.chunk, self.trainer.grad_comm_dtype)
        self.g_opt.replay()
        self.trainer.step_no += 1
        return self.loss
