"""Optimizer side of the measured step, MI355X-first: all trainable parameters and their gradients
live in two flat float32 arenas (one memset to zero the grads, one fused clip+AdamW kernel over
everything, gradient all-reduce over RCCL in a few large chunks instead of per-tensor buckets).

Restates the tail of Seq2SeqCMTAgent.train (VLN-HAMT/finetune_src/r2r/agent_cmt.py:809-832):
zero_grad, backward, clip_grad_norm_(40.), AdamW step; DDP's constructor broadcast and gradient
averaging (:61-63); the three optimizer parameter groups of the shipped "variant4" warm-up
(:82-96, r2r/main.py:202-255); and the optimizer part of the checkpoints (:837-870)."""
import os

import torch
import torch.distributed as dist

from . import _lib, ops


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _exchange():
    """True when the gradient exchange runs: several ranks - or ONE rank with VLNI_FORCE_COLLECTIVES=1, which sends the whole flush ->
    side-stream all-reduce -> graph pipeline through the process group on a one-GPU box (a 1-rank RCCL communicator still launches its
    kernels, keeps its watchdog thread alive beside the graph captures, and completes work objects asynchronously)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("VLNI_FORCE_COLLECTIVES") == "1"


def sync_autotune(src=0):
    """Every rank adopts rank `src`'s kernel choices (GEMM pipelines per shape, weight-gradient variant x split): the ranks time their
    candidates independently, and a different winner on one rank is a different step time under a max-over-ranks clock."""
    if _world() == 1:
        return
    box = [(dict(ops._GEMM_BEST), dict(ops._TN_BEST), dict(ops._TNB_BEST))] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() != src:
        for mine, theirs in zip((ops._GEMM_BEST, ops._TN_BEST, ops._TNB_BEST), box[0]):
            mine.clear()
            mine.update(theirs)


def _all_reduce_sum(t, async_op=False):
    """SUM all-reduce of one contiguous tensor. gloo (CPU tests, 1-GPU rehearsals) takes device tensors through the host."""
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def allreduce_mean_(flat, chunk_elems, comm_dtype=None):
    """In-place mean over the data-parallel group (DDP's gradient averaging, r2r/agent_cmt.py:61-63) on ONE flat
    buffer: a handful of large all-reduces (RCCL over xGMI on the GPU box, gloo in the CPU tests) instead of
    per-tensor buckets. comm_dtype=torch.bfloat16 halves the bytes on the xGMI links (the reduction then runs in bf16,
    like DDP's bf16 compression hook; the arena stays float32). No-op without an initialised process group."""
    ws = _world()
    if ws == 1:
        return flat
    if comm_dtype is None or comm_dtype == flat.dtype:
        works = [_all_reduce_sum(flat[o:o + chunk_elems], async_op=True) for o in range(0, flat.numel(), chunk_elems)]
        for w in works:
            if w is not None:
                w.wait()
        return flat.mul_(1.0 / ws)
    # pre-divide so the bf16 sum of ws terms stays in range, reduce compressed chunks, expand back
    parts = []
    for o in range(0, flat.numel(), chunk_elems):
        c = _pack(flat[o:o + chunk_elems], 1.0 / ws, comm_dtype)
        parts.append((o, c, _all_reduce_sum(c, async_op=True)))
    for o, c, w in parts:
        if w is not None:
            w.wait()
        _unpack(c, flat[o:o + c.numel()])
    return flat


def _pack(src, scale, dtype):
    """(dtype)(src * scale) in ONE kernel on the device (vlni_scale_cast); plain torch on the CPU (gloo tests)."""
    if src.is_cuda and dtype == torch.bfloat16 and src.dtype == torch.float32:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
        _lib.call("vlni_scale_cast", ops.F32, ops.BF16, src.data_ptr(), out.data_ptr(), src.numel(), scale, ops._st())
        return out
    return (src * scale).to(dtype)


def _unpack(c, dst):
    if c.is_cuda and c.dtype == torch.bfloat16 and dst.dtype == torch.float32:
        _lib.call("vlni_scale_cast", ops.BF16, ops.F32, c.data_ptr(), dst.data_ptr(), c.numel(), 1.0, ops._st())
    else:
        dst.copy_(c)


def cut_ranges(units, n, pieces):
    """Element ranges [lo, hi) of an arena of n elements, cut at unit boundaries (units = (first element, elements), e.g. a packed
    q/k/v triple that must stay whole) into about `pieces` ranges of equal size."""
    ends = sorted({u0 + un for u0, un in units if 0 < u0 + un < n})
    cuts = []
    for j in range(1, max(1, pieces)):
        if ends:
            c = min(ends, key=lambda e: abs(e - j * n / pieces))      # the unit boundary nearest to the j-th equal cut
            if not cuts or c > cuts[-1]:
                cuts.append(c)
    edges = [0] + cuts + [n]
    return list(zip(edges[:-1], edges[1:]))


def reduce_range_(flat, lo, hi, ws, chunk_elems, comm_dtype=None):
    """Mean over the ranks of flat[lo:hi], in chunks of chunk_elems, issued on the current stream (the side stream of the
    pipeline). float32 payload: all-reduce in place, then one scaling kernel; bf16 payload: pack (x 1/ws) -> all-reduce -> unpack."""
    for o in range(lo, hi, chunk_elems):
        seg = flat[o:min(hi, o + chunk_elems)]
        if comm_dtype in (None, torch.float32):
            w = _all_reduce_sum(seg, async_op=True)
            if w is not None:
                w.wait()
            if seg.is_cuda:
                _lib.call("vlni_scale_cast", ops.F32, ops.F32, seg.data_ptr(), seg.data_ptr(), seg.numel(), 1.0 / ws, ops._st())
            else:
                seg.mul_(1.0 / ws)
        else:
            c = _pack(seg, 1.0 / ws, comm_dtype)
            w = _all_reduce_sum(c, async_op=True)
            if w is not None:
                w.wait()
            _unpack(c, seg)


class FlatTrainer:
    """clip_grad_norm_ + AdamW over flat arenas, with torch.optim-style parameter groups.

    groups=None: ONE group of ALL model.parameters(), like the reference's `optimizer(self.vln_bert.parameters(), lr)` (r2r/agent_cmt.py:98).
    Otherwise a list of dicts like torch.optim's param_groups - {"params": iterable, "lr": float (default lr), "trainable": bool (default
    True), "name": str}. The position of a parameter in its group's list is its index in state_dict() / load_state_dict(), exactly as in
    torch.optim, so checkpoints travel both ways whether or not some parameters are frozen.

    What goes into the arena: every listed parameter that requires a gradient, plus every parameter of a group declared trainable=False
    (a later stage switches it on with `set_group`; declaring it off also clears requires_grad, like set_group(trainable=False)). A
    parameter the MODEL froze (requires_grad False inside a trainable group: --fix_lang_embedding, update_lang_bert=False) keeps its flag and
    stays outside the arena: it has an index, never a state entry - torch.optim.AdamW skips parameters without a gradient the same way.
    requires_grad is only ever set by set_group() (and by declaring a group off), never switched ON here.

    Per group the learning rate and the trainable flag live in device memory: changing them is a small device write that a captured step
    picks up on its next replay (switching `trainable` also flips requires_grad, which changes the autograd graph: capture again after it).

    Only the parameters of THIS trainer are marked for direct gradient accumulation / deferred weight gradients (ops._direct,
    ops._wb_grad_to look at the marks): other models in the process keep plain autograd accumulation.

    Parameters that never receive a gradient (behind a detach(): the embeddings under --fix_lang_embedding; unused heads) are left alone
    by the step kernel, weight decay included, and have no state entry - like torch.optim.AdamW's grad-None parameters. The rule is per
    ELEMENT (g, m, v all exactly zero), so a row of an embedding table that no batch has touched yet does not decay either, where torch
    decays the whole tensor once any row has a gradient; and a parameter that did receive gradients keeps being updated (its moments
    decay) in a later step in which its gradient happens to be missing, where torch would skip that step for it."""

    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=40.0,
                 chunk_mb=128, grad_comm_dtype=None, groups=None, broadcast=True, overlap_chunks=4,
                 loss_scale=None, growth_interval=0):
        self.model = model
        if groups is None:
            groups = [{"params": list(model.parameters()), "lr": lr, "name": "all"}]
        assert 1 <= len(groups) <= 8, "1..8 parameter groups"
        # arena order, group by group: inside a group the q/k/v projections of every attention module sit back to back (weights,
        # then biases), so the packed [2304,768] QKV gradient is ONE wgrad GEMM into a contiguous view (ops._packed_grad)
        qkv_units = []
        for mod in model.modules():
            if all(hasattr(mod, n) for n in ("query", "key", "value")):
                for attr in ("weight", "bias"):
                    unit = [getattr(getattr(mod, n), attr) for n in ("query", "key", "value")]
                    if all(p is not None for p in unit):
                        qkv_units.append(unit)
        self.groups, self.params, seen = [], [], set()
        self._units = []                    # (first offset, elements) of what a flush chunk / all-reduce chunk must not split
        offs, n = {}, 0
        for gi, g in enumerate(groups):
            listed = list(g["params"])
            assert listed, f"parameter group {gi} is empty"
            assert not ({id(p) for p in listed} & seen), "a parameter appears in two groups"
            seen |= {id(p) for p in listed}
            on = bool(g.get("trainable", True))
            plist = [p for p in listed if p.requires_grad or not on]      # the rest: frozen by the model, index only (class docstring)
            assert plist, f"parameter group {gi} ({g.get('name', gi)}) has no parameter that requires a gradient"
            ids = {id(p) for p in plist}
            order, placed = [], set()
            for unit in qkv_units:
                if all(id(p) in ids for p in unit) and not any(id(p) in placed for p in unit):
                    order.append(unit)
                    placed.update(id(p) for p in unit)
            order += [[p] for p in plist if id(p) not in placed]
            for unit in order:
                u0 = n
                for p in unit:
                    offs[id(p)] = n
                    self.params.append(p)
                    n += (p.numel() + 7) // 8 * 8     # slots aligned to 16 bytes in the 16-bit mirror too (32 B in float32)
                self._units.append((u0, n - u0))
            # which parameters the MODEL had trainable at construction: set_group(trainable=True) switches only those back on (a group declared
            # trainable=False also lists parameters the model froze itself - fix_lang_embedding, update_lang_bert=False - and those stay frozen)
            self.groups.append({"name": g.get("name", f"group{gi}"), "params": plist, "listed": listed, "lr": float(g.get("lr", lr)),
                                "trainable": on, "end": n, "model_trainable": {id(p) for p in plist if p.requires_grad}})
        self._off = offs
        self.n = n
        dev = self.params[0].device
        assert dev.type == "cuda", "FlatTrainer needs the model on the GPU"
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p in self.params:
                o = offs[id(p)]
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p._vlni_direct = True         # kernels accumulate this parameter's gradient straight into the arena
                p._vlni_defer = True          # ... and its weight-gradient GEMMs are queued and grouped (ops.flush_wgrads)
        self.hp = (lr, betas[0], betas[1], eps, weight_decay)
        self.max_norm = max_norm
        self.step_no = 0
        # [0]: sum of squares of the gradients (clip_grad_norm_); [32:]: the partial reduction's spread accumulators (ops.GradArena)
        self._sumsq_all = torch.zeros(32 * (1 + ops.SUMSQ_SLOTS), dtype=torch.float32, device=dev)
        self.sumsq = self._sumsq_all[:1]
        # everything that changes from step to step lives on the device, so a captured step replays correctly
        # (layouts: include/vlni.h, vlni_optim_prepare_groups)
        G = len(self.groups)
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)
        self.gstate = torch.zeros(4 * G, dtype=torch.float32, device=dev)
        self.grp_end = torch.tensor([g["end"] for g in self.groups], dtype=torch.int64, device=dev)
        self.grp_lr = torch.tensor([g["lr"] for g in self.groups] + [1.0 if g["trainable"] else 0.0 for g in self.groups],
                                   dtype=torch.float32, device=dev)
        self.growth_interval = int(growth_interval)
        self.state[4] = float(loss_scale) if loss_scale else 1.0
        for g in self.groups:
            if not g["trainable"]:              # a group declared off: as set_group(trainable=False). Nothing is ever switched ON here.
                for p in g["params"]:
                    p.requires_grad_(False)
        self.chunk = chunk_mb * (1 << 20) // 4
        self.grad_comm_dtype = grad_comm_dtype
        self.overlap_chunks = max(1, int(overlap_chunks))
        self._side = None
        self.time_exchange = False          # True: allreduce_grads records HIP events for exchange_report()
        self.skip_exchange = False          # True: the pipeline runs without the collective itself (what the exchange costs = step - this)
        self.graph_epoch = 0
        # compute-dtype mirror of the parameter arena, kept current by the AdamW kernel (ops.ShadowCache hands out views of it):
        # bfloat16, or float16 when the model computes in float16 (then pair it with loss_scale / growth_interval)
        cd = getattr(model, "compute_dtype", torch.bfloat16)
        self.flat_b = torch.empty(n, dtype=cd if cd in ops.H16 else torch.bfloat16, device=dev)
        if broadcast and _world() > 1:
            self.broadcast_from(0)
        if _exchange():
            # leave a few CUs' worth of every one-round weight-gradient launch to the collective's own kernels: a launch sized for all 256
            # CUs gives RCCL no CU until one of its (100+ us) tiles retires. Costs the flush ~3 % where a launch is exactly one round.
            self._reserve_before = ops.RESERVE_CUS
            ops.RESERVE_CUS = int(os.environ.get("VLNI_RESERVE_CUS", "8"))
        ops.SHADOWS.set_arena(self.flat_p, self.flat_b)
        ops.reserve_staging()               # pinned staging for launch tables a capture may have to upload (ops._dev_table)

    # ---- replicas ---------------------------------------------------------------------------------------------------
    def broadcast_from(self, src=0):
        """What DDP's constructor does (r2r/agent_cmt.py:61-63: ranks are seeded seed+rank, r2r/main.py:446, so anything a
        checkpoint does not cover starts different per rank): rank `src`'s arena, the parameters outside it and the buffers."""
        def bc(t):
            if t.is_cuda and dist.get_backend() == "gloo":
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src)
        with torch.no_grad():
            bc(self.flat_p)
            for p in self.model.parameters():
                if id(p) not in self._off:
                    bc(p.data)
            for b in self.model.buffers():
                bc(b.data)
        ops.SHADOWS.invalidate()

    # ---- groups -----------------------------------------------------------------------------------------------------
    def set_lr(self, lr, group=None):
        """Learning rate of one group (index or name) or of all groups: a device write, picked up by a captured step."""
        for k, g in enumerate(self.groups):
            if group is None or group == k or group == g["name"]:
                g["lr"] = float(lr)
                self.grp_lr[k] = float(lr)

    def set_group(self, group, lr=None, trainable=None):
        k = group if isinstance(group, int) else [g["name"] for g in self.groups].index(group)
        g = self.groups[k]
        if lr is not None:
            self.set_lr(lr, k)
        if trainable is not None and bool(trainable) != g["trainable"]:
            g["trainable"] = bool(trainable)
            self.grp_lr[len(self.groups) + k] = 1.0 if trainable else 0.0
            for p in g["params"]:
                p.requires_grad_(bool(trainable) and id(p) in g["model_trainable"])
            ops.SHADOWS.invalidate()          # shadows of frozen parameters are cached as never-stale
            self.graph_epoch += 1             # the autograd graph changed: a captured step has to be captured again

    @property
    def loss_scale(self):
        """1-element device tensor S: multiply the loss by it before backward() (fp16 runs); the step divides it out again."""
        return self.state[4:5]

    # ---- step -------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        ops._WQ.clear()
        # zero fill of what this step's partial reduction will not store over (ops.GradArena); one rank: the reduction also leaves the
        # stored gradients' sum of squares in self.sumsq
        ops.GRADS.begin(self.flat_g, None if _exchange() else self._sumsq_all)

    def flush(self, lo=None, hi=None, fold=False):
        """Completes the gradient arena (deferred grouped weight-gradient GEMMs), optionally only elements [lo, hi). fold: the caller is the
        optimizer step (or the captured graph in front of it) - nothing touches the arena between this and the norm, so the reduction may
        leave the sum of squares of what it stores in self.sumsq; after any other flush step() takes the norm from the arena itself."""
        if not fold:
            ops.GRADS.fold_ok = False
        if lo is None:
            ops.flush_wgrads()
        else:
            base = self.flat_g.data_ptr()
            ops.flush_wgrads(base + 4 * lo, base + 4 * hi)

    def comm_ranges(self):
        """Element ranges [lo, hi) of the arena: the stages of the overlapped flush -> all-reduce pipeline (cut_ranges)."""
        return cut_ranges(self._units, self.n, self.overlap_chunks)

    def _reduce_range(self, lo, hi, ws):
        if self.skip_exchange:              # measurement only (bench.py `allreduce_ms_exposed`): the step without its collective - every rank keeps its
            return                          # own gradients; never set in training
        reduce_range_(self.flat_g, lo, hi, ws, self.chunk, self.grad_comm_dtype)

    def allreduce_grads(self, flushers=None):
        """Flushes the deferred weight gradients chunk by chunk (arena order) and averages each chunk over the ranks on a side
        stream as soon as it is complete: the all-reduce of chunk k runs under the weight-gradient GEMMs of chunk k + 1
        (DDP overlaps its buckets with backward the same way, r2r/agent_cmt.py:61-63,827). flushers: per-range callables
        (captured graphs) instead of eager launches."""
        ws = _world()
        if not _exchange():
            if flushers is None:
                self.flush()
            else:
                for f in flushers:
                    f()
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        ranges = self.comm_ranges()
        timed = self.time_exchange
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for k, (lo, hi) in enumerate(ranges):
            if flushers is None:
                self.flush(lo, hi)
            else:
                flushers[k]()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                if timed and k == 0:
                    ev[0].record()
                self._reduce_range(lo, hi, ws)
        if timed:
            ev[1].record(main)                 # the last flush is done: from here on the main stream only waits for the exchange
            ev[2].record(self._side)
            self._exchange_events = ev
        main.wait_stream(self._side)

    def exchange_report(self):
        """After a step with time_exchange = True: what the gradient exchange cost (HIP events; call after a synchronize)."""
        ev = getattr(self, "_exchange_events", None)
        if ev is None:
            return None
        ranges = self.comm_ranges()
        es = 2 if self.grad_comm_dtype in (torch.bfloat16, torch.float16) else 4
        return {"exposed_ms": round(max(0.0, ev[1].elapsed_time(ev[2])), 3),      # exchange still running after the last flush kernel
                "exchange_span_ms": round(ev[0].elapsed_time(ev[2]), 3),           # first range's pack -> last range's unpack
                "ranges": len(ranges), "payload_bytes_per_range": [(hi - lo) * es for lo, hi in ranges],
                "payload_dtype": str(self.grad_comm_dtype or torch.float32).replace("torch.", "")}

    def step(self):
        self.flush(fold=True)
        st = ops._st()
        self.step_no += 1
        _, b1, b2, eps, wd = self.hp
        G = len(self.groups)
        mine = ops.GRADS.arena is not None and ops.GRADS.arena.data_ptr() == self.flat_g.data_ptr()
        if not (mine and ops.GRADS.finish_sumsq()):
            self.sumsq.zero_()
            _lib.call("vlni_sumsq", self.flat_g.data_ptr(), self.n, self.sumsq.data_ptr(), st)
        if mine:
            ops.GRADS.end()
        _lib.call("vlni_optim_prepare_groups", self.sumsq.data_ptr(), self.max_norm, b1, b2, self.state.data_ptr(),
                  self.gstate.data_ptr(), self.grp_lr.data_ptr(), G, self.growth_interval, st)
        _lib.call("vlni_adamw_step_groups", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                  self.flat_b.data_ptr(), ops._DT[self.flat_b.dtype], self.n, self.grp_end.data_ptr(), self.grp_lr.data_ptr(), self.gstate.data_ptr(),
                  G, b1, b2, eps, wd, self.state.data_ptr(), st)
        ops.SHADOWS.invalidate(optimizer_step=ops.SHADOWS.arena is not None and ops.SHADOWS.arena[0] is self.flat_p)

    def grad_norm(self):
        """Total gradient norm the last step clipped against (unscaled)."""
        return float(self.state[7])

    # ---- checkpoints (the 'optimizer' entry of r2r/agent_cmt.py:837-870) -------------------------------------------------
    def state_dict(self):
        """torch.optim.AdamW's layout: param_groups[k]["params"] = the indices of the group's parameters in the order they were given
        (parameters the model froze included), state[i] = {step, exp_avg, exp_avg_sq} for the parameters that are optimised."""
        _, b1, b2, eps, wd = self.hp
        steps = self.gstate.view(-1, 4)[:, 0].tolist()
        state, pgs, i = {}, [], 0
        for k, g in enumerate(self.groups):
            ids = []
            for p in g["listed"]:
                o = self._off.get(id(p))
                # no entry for a parameter that never received a gradient (all-zero moments: the step kernel left it alone), as in torch
                if o is not None and steps[k] > 0 and bool(self.v[o:o + p.numel()].any()):
                    state[i] = {"step": torch.tensor(float(steps[k])), "exp_avg": self.m[o:o + p.numel()].view(p.shape).clone(),
                                "exp_avg_sq": self.v[o:o + p.numel()].view(p.shape).clone()}
                ids.append(i)
                i += 1
            pgs.append({"lr": g["lr"], "betas": (b1, b2), "eps": eps, "weight_decay": wd, "name": g["name"],
                        "trainable": g["trainable"], "params": ids})
        return {"state": state, "param_groups": pgs,
                "vlni": {"step_no": self.step_no, "loss_scale": float(self.state[4]), "good_steps": float(self.state[6])}}

    def load_state_dict(self, sd):
        """Takes this class's own state_dict() and torch.optim.AdamW's (e.g. the 'optimizer' entry of a reference checkpoint, r2r/agent_cmt.py:
        837-870): indices count the parameters of a group in the order they were given; entries of parameters outside the arena (frozen by
        the model) are ignored, missing entries start from zero moments."""
        G = len(self.groups)
        assert len(sd["param_groups"]) == G, "optimizer state has a different number of parameter groups"
        gs = torch.zeros(G, 4)
        _, b1, b2, _, _ = self.hp
        with torch.no_grad():
            for k, (g, pg) in enumerate(zip(self.groups, sd["param_groups"])):
                assert len(pg["params"]) == len(g["listed"]), \
                    f"group {k}: the state lists {len(pg['params'])} parameters, this trainer's group has {len(g['listed'])}"
                self.set_group(k, lr=pg["lr"], trainable=pg.get("trainable", g["trainable"]))
                t = 0.0
                for p, i in zip(g["listed"], pg["params"]):
                    o = self._off.get(id(p))
                    if o is None:
                        continue
                    st = sd["state"].get(i)
                    if st is not None:
                        self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                        self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                        t = max(t, float(st["step"]))
                    else:
                        self.m[o:o + p.numel()].zero_()
                        self.v[o:o + p.numel()].zero_()
                gs[k] = torch.tensor([t, 1.0 - b1 ** t, 1.0 - b2 ** t, 0.0])
            self.gstate.copy_(gs.reshape(-1))
            extra = sd.get("vlni", {})
            self.step_no = int(extra.get("step_no", int(gs[:, 0].max())))
            self.state[3] = float(self.step_no)
            self.state[4] = float(extra.get("loss_scale", 1.0))
            self.state[6] = float(extra.get("good_steps", 0.0))

    def close(self):
        """Takes the marks off this trainer's parameters (direct gradient accumulation, deferred weight gradients) and drops
        the registered parameter arena / dropout seed base. The parameters keep pointing into the arenas."""
        ops._WQ.clear()
        ops._PART_BUFS.clear()                 # row-split workspaces and reduction tables of this arena's gradients
        ops._PART_TABLES.clear()
        ops._KEEPALIVE.clear()                 # (replaced ones that older captured graphs of this trainer still pointed at)
        if ops.GRADS.arena is not None and ops.GRADS.arena.data_ptr() == self.flat_g.data_ptr():
            ops.GRADS.reset()
        for p in self.params:
            p._vlni_direct = p._vlni_defer = p._vlni_queued = p._vlni_added = False
        if ops.SHADOWS.arena is not None and ops.SHADOWS.arena[0] is self.flat_p:
            ops.SHADOWS.set_arena(None, None)
        ops.set_seed_base(None)
        if hasattr(self, "_reserve_before"):   # the CU reservation for a gradient exchange was this trainer's: back to what it found
            ops.RESERVE_CUS = self._reserve_before
            del self._reserve_before

    def __del__(self):
        # a trainer dropped without close(): the module-global arena bookkeeping must not keep its gradient arena alive
        try:
            if ops.GRADS.arena is not None and ops.GRADS.arena.data_ptr() == self.flat_g.data_ptr():
                ops.GRADS.reset()
        except Exception:
            pass

    def set_defer(self, on):
        """Deferred (grouped, one launch per parameter and episode) vs immediate weight-gradient GEMMs."""
        for p in self.params:
            p._vlni_defer = bool(on)

    def capture(self, fwd_bwd, warmup=1, stages=()):
        """Captures the training step into hipGraphs and returns a callable that replays them.

        fwd_bwd() runs forward + backward (on fixed-shape, fixed-address inputs; refill them in place between calls) and
        returns the loss tensor. One rank: graph 1 = zero_grad + fwd_bwd + deferred weight-gradient flush (including the
        compute-dtype shadow refresh of every parameter), graph 2 = clip + AdamW. Several ranks: the flush is cut into
        `overlap_chunks` graphs and the RCCL all-reduce of each chunk runs eagerly on a side stream under the next chunk's
        weight-gradient GEMMs (a collective is never part of a capture). `warmup` REAL steps run first on a side stream
        (lazy initialisation, GEMM autotune). Drop every reference to earlier losses / outputs before calling this: an
        autograd graph that is still alive keeps its AccumulateGrad nodes on the old stream, which breaks the capture.
        `stages`: phases to capture as separate graphs in front of fwd_bwd (GraphedStep)."""
        return GraphedStep(self, fwd_bwd, warmup, stages)


class GraphedStep:
    """One training step as replayable hipGraphs: [zero_grad + forward + backward] -> gradient flush (one graph per exchange range
    when world > 1) -> [clip + AdamW]. `stages`: callables that run BEFORE fwd_bwd, each captured as its own graph in the same
    memory pool - the phases of an episode between which the host must act (a sampled rollout: begin / step 0 / ... / step T-1 with the
    action choice and the simulator in between, hamt.episode.TapedEpisode); replay them in order with stage(i), then finish().
    Calling the object replays everything back to back."""

    def __init__(self, trainer, fwd_bwd, warmup=1, stages=()):
        import gc
        self.trainer = trainer
        self.epoch = trainer.graph_epoch
        stages = list(stages)
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                trainer.zero_grad()
                for st in stages:
                    st()
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
        split = _exchange()
        # With a process group alive its watchdog thread may make HIP calls while this thread captures: capture in thread-local error
        # mode so that those calls neither fail nor invalidate the capture (torch's default "global" mode would do both)
        mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
        self.g_stages, pool = [], None

        def first():
            self.seed_base.add_(7919)
            trainer.zero_grad()
        for i, st in enumerate(stages):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **(dict(pool=pool) if pool is not None else {}), **mode):
                if i == 0:
                    first()
                st()
            pool = g.pool()
            self.g_stages.append(g)
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb, **(dict(pool=pool) if pool is not None else {}), **mode):
            if not stages:
                first()
            self.loss = fwd_bwd()
            if not split:
                trainer.flush(fold=True)
        self.g_flush = []
        if split:
            base = trainer.flat_g.data_ptr()
            for lo, hi in trainer.comm_ranges():
                if not any(base + 4 * lo <= k < base + 4 * hi for k in list(ops._WQ) + [base + 4 * o for o in ops.GRADS.pending]):
                    self.g_flush.append(None)              # nothing deferred in this range (small parameters only)
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.g_fb.pool(), **mode):
                    trainer.flush(lo, hi)
                self.g_flush.append(g)
            assert not ops._WQ, "a queued weight gradient lies outside every flush range"
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool(), **mode):
            trainer.step()
        trainer.step_no -= 1                   # recorded, not executed

    def stage(self, i):
        if self.epoch != self.trainer.graph_epoch:
            raise RuntimeError("a parameter group was switched on / off after this step was captured: capture() again")
        self.g_stages[i].replay()

    def finish(self):
        if self.epoch != self.trainer.graph_epoch:
            raise RuntimeError("a parameter group was switched on / off after this step was captured: capture() again")
        self.g_fb.replay()
        if self.g_flush:
            self.trainer.allreduce_grads(flushers=[(g.replay if g is not None else (lambda: None)) for g in self.g_flush])
        self.g_opt.replay()
        self.trainer.step_no += 1
        return self.loss

    def __call__(self):
        for i in range(len(self.g_stages)):
            self.stage(i)
        return self.finish()
