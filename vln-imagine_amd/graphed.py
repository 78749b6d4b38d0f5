"""One autograd node per wrapper call (round 6): what an UNCHANGED reference agent gets from `VLNBertCMT` / `VLNBert`.

An eager `visual` call of the drop-in modules is ~45 autograd nodes and ~110 C calls, and its backward the same again on the autograd thread:
at the bench's shapes an agent iteration (Seq2SeqCMTAgent.rollout + loss.backward(), r2r/agent_cmt.py:400-700,827-832) was bound by that
host work (40-73 ms depending on the box's CPU) against ~30 ms of kernels. Here a call of mode M with inputs of signature S (shapes, dtypes,
which inputs need gradients), the k-th such call since the last backward pass, is ONE `torch.autograd.Function`:

  forward  = copy the inputs into the entry's static buffers, replay its captured FORWARD hipGraph, hand out its static outputs;
  backward = copy the output gradients in, replay its captured BACKWARD hipGraph, hand out the static input gradients.

The graphs are captured from the very same model code (`NavCMT.forward`) with autograd on: the backward graph is `torch.autograd.grad`
of the captured forward's outputs, recorded right behind it. Inside it the operators do
what they do in an agent's eager backward (ops.GradSession): LayerNorm / embedding / bias gradients accumulate in the kernels into the
session's persistent flat buffer, and the projections QUEUE their (dY, X) pairs - the grouped weight-gradient launches over all steps
still run once, at the end of the agent's `loss.backward()`. What is Python-side state in that protocol is replayed by hand after each
backward replay: the queue entries (static buffers, so the same pairs every iteration) and the "touched" marks that decide which parameters
keep a gradient.

Entries are per OCCURRENCE (the k-th call of a signature since the last backward) because the saved activations of step t must survive
until its backward: T steps of an episode are T entries with their own graph memory pools. The first time a (signature, k) is seen the call
runs eagerly (which also times the GEMM pipelines for its shapes); from the second time on it is captured and replayed. Shapes that keep
changing (ragged real batches) therefore simply stay on the eager path; at most `MAX_ENTRIES` entries are kept.

Dropout: the in-kernel masks are hash(seed + *base, element) with the seed a launch argument (a constant of the graph) and `base` a device
counter (ops.set_seed_base) that this module advances at the first call after a backward pass - forward and backward replays of one iteration read the same
value, the next iteration draws new masks. torch's own dropout (feature dropout of the wrappers) is graph-safe by itself (philox offsets).
"""
import os

import torch

from . import ops

ENABLED = os.environ.get("VLNI_GRAPHED_MODES", "1") == "1"
MAX_ENTRIES = int(os.environ.get("VLNI_GRAPHED_MAX", "96"))
MAX_OCCURRENCE = 48


class _Entry:
    __slots__ = ("sig", "fn", "pool", "fwd", "bwd", "ins", "ins_w", "aliased", "in_req", "outs", "live_outs", "out_req", "gouts", "gins", "queued",
                 "touched", "seen", "used", "broken", "session")

    def __init__(self, sig, fn):
        self.sig, self.fn = sig, fn
        self.fwd = self.bwd = None
        self.outs = None
        self.seen, self.used, self.broken = 0, 0, False


class _Call(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, anchor, *ins):
        ctx.entry = entry
        ctx.set_materialize_grads(False)
        for s, t in zip(entry.ins_w, ins):               # (.data aliases: the captured graph's saved inputs keep their version counters)
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        entry.fwd.replay()
        outs = tuple(o.detach() for o in entry.outs)
        ctx.mark_non_differentiable(*[o for o, r in zip(outs, entry.out_req) if not r])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        e = ctx.entry
        ses = e.session
        with torch.no_grad():
            _open(ses)
            for s, g in zip(e.gouts, gouts):
                if s is None:
                    continue
                if g is None:
                    s.zero_()
                elif s.data_ptr() != g.data_ptr():
                    s.copy_(g)
            e.bwd.replay()
            # the Python half of the operators' backward protocol (ops.GradSession), replayed by hand
            ep = ses.epoch
            for p in e.touched:
                p._vlni_touch = ep
            for key, (wv, bv, pairs, prms) in e.queued.items():
                ent = ses.queue.get(key)
                if ent is None:
                    ent = ses.queue[key] = (wv, bv, [])
                    for prm in prms:
                        prm._vlni_queued = True
                ent[2].extend(pairs)
        return (None, None) + tuple(g.detach() if g is not None else None for g in e.gins)


def _open(ses):
    """The agent model's GradSession for the backward pass in flight (opened the way the operators open it, ops._session)."""
    if ses.active and ses.task != torch._C._current_graph_task_id() and not ses.hold:
        ses.queue.clear()                                  # left behind by a backward pass that raised
        ses.end()
    if not ses.active:
        ses.begin()



def _capture_forward(e, ins, g):
    e.pool = torch.cuda.graph_pool_handle()
    e.ins, e.aliased = [], []
    for t in ins:
        if t.data_ptr() in g.static_ptrs:                 # another entry's static output (txt_embeds of the `language` call): read in place
            e.ins.append(t.detach().requires_grad_(t.requires_grad))
            e.aliased.append(True)
        else:
            e.ins.append(t.detach().clone().requires_grad_(t.requires_grad))
            e.aliased.append(False)
    e.in_req = [t.requires_grad for t in ins]
    e.ins_w = [t.data for t in e.ins]
    torch.cuda.synchronize()
    e.fwd = torch.cuda.CUDAGraph()
    with torch.cuda.graph(e.fwd, pool=e.pool, stream=g.stream):
        with torch.enable_grad():
            outs = e.fn(*e.ins)
    outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
    e.live_outs = outs                                    # with their graph: the backward capture differentiates them
    e.outs = [o.detach() for o in outs]
    e.out_req = [o.requires_grad for o in outs]
    for o in e.outs:
        g.static_ptrs.add(o.data_ptr())


def _capture_backward(e, ses):
    """Recorded right behind the forward capture, on the calling thread (a capture on the autograd engine's worker thread, the first time the
    entry's backward runs, ended in a segmentation fault inside capture_end on this stack): every differentiable output is given a gradient
    buffer; one that receives no gradient in a backward pass is zero-filled there. The gradient session is opened for the capture alone."""
    e.gouts = [torch.zeros_like(o) if r else None for o, r in zip(e.outs, e.out_req)]
    assert not ses.active, "vln_imagine_amd.graphed: a wrapper call inside a backward pass"
    ses.begin(register=False)
    try:
        _capture_backward_open(e, ses)
    finally:
        ses.end(quiet=True)


def _capture_backward_open(e, ses):
    diff_outs = [o for o, s in zip(e.live_outs, e.gouts) if s is not None]
    diff_ins = [t for t, r in zip(e.ins, e.in_req) if r]
    # EVERY trainable parameter is named as an input of the differentiation: a node whose only differentiable inputs are parameters (the first
    # projection of precomputed features) lies on no path to the call's tensor inputs and would not run at all otherwise. The operators return
    # None for what their kernels accumulated in place; a gradient that arrives through a torch-native node instead (a slice of an embedding
    # table) is added onto the parameter's gradient inside the captured graph.
    prms = [p for p in ses.params if p.requires_grad]
    before = {k: len(v[2]) for k, v in ses.queue.items()}
    ses.recorder, ses.hold = [], True
    native = []
    torch.cuda.synchronize()
    e.bwd = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(e.bwd, pool=e.pool, stream=ses._graphed.stream):
            if diff_outs:
                with torch.enable_grad():
                    gi = torch.autograd.grad(diff_outs, diff_ins + prms, [s for s in e.gouts if s is not None], allow_unused=True)
                for p_, g_ in zip(prms, gi[len(diff_ins):]):
                    if g_ is not None:
                        if p_.grad is None:
                            raise RuntimeError("vln_imagine_amd.graphed: parameter without a session gradient inside a capture")
                        p_.grad.add_(g_.to(p_.grad.dtype).view_as(p_.grad))
                        native.append(p_)
            else:
                gi = ()
    finally:
        rec, ses.recorder, ses.hold = ses.recorder, None, False
    gi = list(gi[:len(diff_ins)])
    e.gins = [gi.pop(0) if r else None for r in e.in_req] if diff_outs else [None] * len(e.in_req)
    e.live_outs = None                                    # the autograd graph of the capture is not needed again (its buffers stay in the pool)
    seen, e.touched = set(), []
    for p in list(rec) + native:
        if p is not None and id(p) not in seen:
            seen.add(id(p))
            e.touched.append(p)
    # what the captured backward queued: during the capture nothing ran, so the entries are taken OUT of the queue again here and re-added
    # after every replay (also this first one)
    e.queued = {}
    for k, (wv, bv, pairs) in list(ses.queue.items()):
        n0 = before.get(k, 0)
        if len(pairs) > n0:
            prms = [p for p in e.touched if p.grad is not None and (p.grad.data_ptr() == wv.data_ptr() or p.grad.data_ptr() == bv.data_ptr())]
            e.queued[k] = (wv, bv, list(pairs[n0:]), prms)
            del pairs[n0:]
            if not pairs and k not in before:
                del ses.queue[k]


class _NoScope:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_SCOPE = _NoScope()


class _Scope:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        g = self.g
        g.depth += 1
        if g.depth > 1:
            return self
        if g.stream is None:
            g.stream = torch.cuda.Stream()
        self.outer = torch.cuda.current_stream()
        g.stream.wait_stream(self.outer)
        self.ctx = torch.cuda.stream(g.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        g = self.g
        g.depth -= 1
        if g.depth == 0:
            self.ctx.__exit__(*a)
            self.outer.wait_stream(g.stream)
        return False


class ModeGraphs:
    """Per agent model: the (mode, signature, occurrence) -> captured entry table."""

    def __init__(self, model):
        self.model = model
        self.entries, self.counts = {}, {}
        self.static_ptrs = set()
        self.seed_base = None
        self.anchor = None
        self.iteration = 0
        self.in_place = None
        self._ids = None
        self.stream, self.depth = None, 0
        self.epoch = -1               # the gradient session's epoch at the last call: a backward pass in between starts a new iteration
        self.capturing = False        # inside a forward capture (model code that shares per-episode state between calls builds its own copy then)
        self.session = None
        self.stats = {"eager": 0, "replayed": 0, "captured": 0}

    def scope(self):
        """`with graphs.scope():` around EVERY model call of the wrapper (captured or not). All of them run on ONE side stream, which is also the
        capture stream: a parameter's AccumulateGrad node keeps the stream it was created on for as long as any autograd graph refers to it,
        and a few parameters (rows of the type / position embedding tables, the history [CLS] token) receive their gradient through torch-native
        nodes - inside a backward capture a node on the legacy default stream would pull that stream into the capture (on this stack: a
        segmentation fault in capture_end). The caller's stream waits for the side stream on exit and vice versa on entry."""
        return _Scope(self) if self._ready() is not None else _NO_SCOPE

    def _ready(self):
        ses = getattr(next(self.model.parameters()), "_vlni_auto", None)
        if ses is None or not ops.AUTO_DEFER or not ENABLED:
            return None
        if not (torch.is_grad_enabled() and self.model.training) or torch.cuda.is_current_stream_capturing() or ops._TAPE is not None:
            return None
        if getattr(ses.params[0], "_vlni_direct", False) and not ses.active:
            return None                                   # a FlatTrainer owns these parameters
        if getattr(self.model, "compute_dtype", None) not in ops.H16:
            return None                                   # the float32 parity path rebuilds packed weight copies by allocation: eager only
        ses._graphed = self
        self.session = ses
        return ses

    def call(self, mode, consts, fn, tensors):
        """fn(*tensors) -> tensor or tuple of tensors, a pure function of its tensor arguments and the model's parameters (no host reads of
        tensor VALUES, no Python state that differs between calls with equal `consts`). Returns fn's outputs."""
        ses = self._ready()
        if ses is None or not tensors or any(not t.is_cuda for t in tensors):
            return fn(*tensors)
        if ses.epoch != self.epoch:
            # first call since a backward pass (the session's epoch moved): occurrences count from 0 again, the in-kernel dropout draws new masks
            # (forward and backward replays of one iteration read the same base), and the weights' 16-bit copies follow the optimizer IN PLACE
            self.epoch = ses.epoch
            self.iteration += 1
            self.in_place = None
            self.counts.clear()
            if self.seed_base is not None:
                self.seed_base.add_(1)
            if self.entries and not ops.SHADOWS.current_for_replay(self.param_ids(ses)):
                self.reset()
        sig = (mode, consts, tuple((tuple(t.shape), t.dtype, t.requires_grad) for t in tensors))
        k = self.counts.get(sig, 0)
        self.counts[sig] = k + 1
        e = self.entries.get((sig, k))
        if e is None:
            if len(self.entries) >= MAX_ENTRIES:
                # full: entries no rollout has used for a while (a signature the data has moved away from) make room - never one of this iteration
                old = [key for key, v in self.entries.items() if v.used < self.iteration - 8]
                for key in sorted(old, key=lambda q: self.entries[q].used)[:max(1, len(self.entries) // 4)]:
                    for o in self.entries[key].outs or ():
                        self.static_ptrs.discard(o.data_ptr())
                    del self.entries[key]
            if k >= MAX_OCCURRENCE or len(self.entries) >= MAX_ENTRIES:
                self.stats["eager"] += 1
                return fn(*tensors)
            e = self.entries[(sig, k)] = _Entry(sig, fn)
        e.seen += 1
        e.used = self.iteration
        e.fn = fn
        if self.in_place is None:                         # once per iteration (nobody swaps gradient tensors in the middle of a rollout)
            self.in_place = self._grads_in_place(ses)
        if e.seen == 1 or e.broken or not self.in_place:
            self.stats["eager"] += 1
            return fn(*tensors)                           # first sight: eager (times the GEMM pipelines of these shapes, forward and backward)
        if e.fwd is None:
            if self.seed_base is None and ops._SEED_BASE[0] is None:
                self.seed_base = torch.zeros(1, dtype=torch.int32, device=tensors[0].device)
                ops.set_seed_base(self.seed_base)
            try:
                e.session = ses
                self.capturing = True
                _capture_forward(e, tensors, self)
                _capture_backward(e, ses)
                self.epoch = ses.epoch                    # (the capture's own session moved it)
                self.stats["captured"] += 1
            except Exception as err:
                # an operator that cannot be captured (a host read of a tensor value, an allocation the capture forbids): this signature stays on the
                # eager path for good; the agent's iteration goes on
                import warnings
                e.broken, e.fwd, e.bwd = True, None, None
                self.epoch = ses.epoch
                self.stats["eager"] += 1
                warnings.warn(f"vln_imagine_amd.graphed: mode {mode!r} could not be captured ({type(err).__name__}: {err}); its calls stay eager")
                return fn(*tensors)
            finally:
                self.capturing = False
        if any(a and s_.data_ptr() != t.data_ptr() for a, s_, t in zip(e.aliased, e.ins, tensors)):
            # an input that was another entry's static output when this entry was captured is a different tensor now: copying into that buffer
            # would overwrite the other entry's output under its consumers - this call runs eagerly
            self.stats["eager"] += 1
            return fn(*tensors)
        self.stats["replayed"] += 1
        if self.anchor is None:
            self.anchor = torch.zeros(1, device=tensors[0].device, requires_grad=True)   # makes the node differentiable when no INPUT needs a gradient
        outs = _Call.apply(e, self.anchor, *tensors)
        return outs if len(outs) > 1 else outs[0]

    def param_ids(self, ses):
        if self._ids is None or self._ids[0] != len(ses.params):
            self._ids = (len(ses.params), {id(p) for p in ses.params})
        return self._ids[1]

    def _grads_in_place(self, ses):
        """The backward graphs accumulate into the session's flat buffer at fixed addresses: every parameter's .grad must be absent (the session
        hands out its view) or already that view. An agent that installs its own gradient tensors stays on the eager path."""
        if ses.flat is None:
            return True
        base = ses.flat.untyped_storage().data_ptr()
        return all(p.grad is None or p.grad.untyped_storage().data_ptr() == base for p in ses.params)

    def reset(self):
        self.entries.clear()
        self.counts.clear()
        self.static_ptrs.clear()


# ---- shape buckets of the wrapper calls -------------------------------------------------------------------------------------------
# Real batches are ragged: the text length, the candidate count and DUET's map size are padded to the batch maximum by the agents
# (agent_cmt.py:130-176, agent.py:98-207) and change from batch to batch. The wrappers pad them up to the next multiple of a bucket size with
# the agents' own padding values (pad ids / False masks / zero features: padded keys are masked, padded candidates score -inf) before the call
# and slice the outputs back, so that a handful of signatures covers the data and their graphs get replayed. Results equal the unpadded
# call's to summation-order rounding (the same property the reference's own batches have). VLNI_GRAPHED_BUCKETS = "text,candidates,map nodes"
# multiples (0 = that dimension as given).
BUCKETS = tuple(int(v) for v in os.environ.get("VLNI_GRAPHED_BUCKETS", "16,4,8").split(","))


def bucket(n, m):
    return n if m <= 1 else -(-n // m) * m


def pad_dim(t, dim, to, value=0):
    """t padded with `value` along `dim` up to length `to` (None stays None; no copy when nothing is to pad)."""
    if t is None or t.shape[dim] >= to:
        return t
    shape = list(t.shape)
    shape[dim] = to - t.shape[dim]
    return torch.cat([t, t.new_full(shape, value)], dim)


def of(model):
    g = getattr(model, "_vlni_mode_graphs", None)
    if g is None:
        g = ModeGraphs(model)
        object.__setattr__(model, "_vlni_mode_graphs", g)
    return g
