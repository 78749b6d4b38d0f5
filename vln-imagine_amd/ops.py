"""Autograd operators of the HAMT/DUET hot path on top of the vlni C-ABI.

Every operator launches hand-written HIP kernels (libvlni.so) on torch's current stream;
torch supplies device memory, the autograd graph between operators, and nothing else.
There is no eager/CPU fallback: a CPU tensor or a missing library raises.

Granularity is one autograd node per transformer SUBLAYER (attention block, FFN block,
bidirectional cross-attention block) so that the backward pass is an explicit kernel
sequence with fused epilogues (bias, GELU / GELU', residual) instead of ~40 tiny nodes.
"""
import bisect
import ctypes
import os
import weakref

import numpy as np

import torch

from . import _lib

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
H16 = (torch.bfloat16, torch.float16)       # the two throughput dtypes: same kernels, bf16 or f16 MFMA (csrc/*_impl.inc)
NEG_MASK = -10000.0


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"vlni ops take float32, bfloat16 or float16 activations, got {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _st():
    """hipStream_t of torch's current stream (the raw getter is ~20x cheaper than torch.cuda.current_stream())."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"vlni: {name} is on {t.device}; the HIP path has no CPU fallback")
    return t


def _rows(t):
    """[..., H] -> contiguous 2-D view [rows, H]."""
    t = t.reshape(-1, t.shape[-1])
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


# =====================================================================================
#  episode tape: step-by-step forward, ONE episode-batched backward
# =====================================================================================
# The reference agent rolls an episode out step by step (the action taken at step t decides the observation of step t + 1) and runs ONE
# backward at the end (r2r/agent_cmt.py:814-827). No output of a `visual` / `history` call feeds a later call's transformer input (history
# tokens are re-encoded from features, vilmodel_cmt.py:1056-1205), so the T calls are the T batch slices of one call on T x B samples:
#   * record(key, t):  the step's forward runs under no_grad; every activation an operator allocates is slice t of an episode-wide
#                      buffer (rows [t B S, (t + 1) B S) of the tensor the batched call would have produced);
#   * ghost(key):      the SAME model code is called once on the T x B inputs; the operators launch nothing and hand out the filled
#                      buffers, so autograd records the batched graph at no kernel cost;
#   * backward:        runs on 6 x longer launches (M ~ 48 k rows: the 256 x 256 GEMM kernel's shapes, one attention / LayerNorm launch per
#                      layer instead of T).
# Dropout: masks are hash(seed, element index) with the sample outermost in every index, and the hash starts with idx * K + seed, so the
# step-t launch of a tensor of E elements per step uses seed + t E K: exactly the window of the batched tensor's mask that the batched
# backward regenerates from the unshifted seed. Seeds are drawn once per operator call and replayed for the later steps and the ghost pass.
_TAPE = None
_HASH_MUL = 0x9E3779B1


class EpisodeTape:
    def __init__(self, T):
        self.T = T
        self.bufs, self.seeds, self.steps = {}, {}, {}
        self.key = self.mode = self.batch_n = None
        self.t = self.i = self.si = 0
        self._open = {}                 # key -> (allocation index, seed index) of the keys of the context in flight that are not the active one

    def reset(self):
        """New episode: fresh dropout seeds; the buffers are kept (a captured step graph replays into the same addresses)."""
        self.seeds, self.steps = {}, {}

    def _enter(self, keys, mode, t, batch_n=None):
        global _TAPE
        assert _TAPE is None, "episode tapes do not nest"
        self.key, self.mode, self.t, self.i, self.si = keys[0], mode, t, 0, 0
        self.batch_n = batch_n          # record mode only: ONE call on the samples of steps 0 .. batch_n - 1 fills the whole buffers
        self._open = {k: (0, 0) for k in keys}
        for k in keys:
            self.bufs.setdefault(k, [])
            self.seeds.setdefault(k, [])
        _TAPE = self

    def _swap(self, key):
        """Makes `key` (one of the keys of the context in flight) the key that allocations and seed draws go to; returns the previous one."""
        prev = self.key
        if key is None or key == prev:
            return prev
        assert key in self._open, f"tape: {key!r} is not open (open: {sorted(self._open)})"
        self._open[prev] = (self.i, self.si)
        self.key = key
        self.i, self.si = self._open[key]
        return prev

    def use(self, key):
        """Context manager: inside, activations and dropout seeds belong to `key`. A recorded step that runs two calls of the model in
        lockstep (the `visual` call and the `history` call of a HAMT step share multi-problem GEMM launches) opens both keys with
        record((k0, k1), t) and switches per allocation, so each key's sequence is the one its own ghost pass will ask for."""
        return _TapeUse(self, key)

    def _exit(self):
        global _TAPE
        self._open[self.key] = (self.i, self.si)
        counts = {}
        for key, (n_alloc, n_seed) in self._open.items():
            if self.mode == "record":
                if self.batch_n:
                    self.steps[key] = self.batch_n
                else:
                    self.steps[key] = self.t + 1 if self.t == 0 else max(self.steps.get(key, 0), self.t + 1)
                if self.t == 0 and n_alloc < len(self.bufs[key]):        # another program than the last episode's (e.g. NavCMT.visual_lang_rows
                    _KEEPALIVE.extend(self.bufs[key][n_alloc:])          # switched, train() -> eval()): step 0 defines it; an older capture may still use the rest
                    del self.bufs[key][n_alloc:]
            counts[key] = (n_alloc, n_seed)
        _TAPE = None
        self.key = self.mode = None
        self._open = {}
        return counts

    def record(self, key, t):
        """key: one tape key, or a tuple of keys recorded by ONE lockstep pass (see use())."""
        return _TapeCtx(self, key if isinstance(key, tuple) else (key,), "record", t)

    def record_steps(self, key, n):
        """RECORD the calls of steps 0 .. n - 1 of `key` as ONE call on n x B samples (teacher forcing: the inputs of all steps are known -
        the history tokens of a HAMT episode are encoded from features alone, vilmodel_cmt.py:576-618): the operators compute, with fresh
        dropout seeds, straight into the episode-wide buffers - exactly the tensors the n per-step calls would have filled (the masks too:
        a step's launch draws the window of the batched tensor's mask, see _shift). The ghost pass of `key` is then this very call again."""
        return _TapeCtx(self, (key,), "record", 0, batch_n=n)

    def ghost(self, key, compute=False):
        """compute=True: the batched call computes for real with the recorded dropout seeds (the reference the tests hold the tape to)."""
        return _TapeCtx(self, (key,), "compute" if compute else "ghost", 0)

    def take(self, shape, dtype, device):
        if self.mode == "compute":
            return torch.empty(shape, dtype=dtype, device=device)
        bufs, i = self.bufs[self.key], self.i
        self.i += 1
        if self.mode == "record" and self.batch_n:
            n = self.batch_n
            assert shape[0] % n == 0, f"tape {self.key!r}: batched allocation {i} of {tuple(shape)} does not split into {n} steps"
            per = shape[0] // n
            full = (self.T * per,) + tuple(shape[1:])
            if i == len(bufs):
                bufs.append(torch.empty(full, dtype=dtype, device=device))
            b = bufs[i]
            if tuple(b.shape) != full or b.dtype != dtype:
                _KEEPALIVE.append(b)                    # an older capture may still write here
                b = bufs[i] = torch.empty(full, dtype=dtype, device=device)
            return b[:n * per]
        if self.mode == "record":
            full = (self.T * shape[0],) + tuple(shape[1:])
            if i == len(bufs):
                assert self.t == 0, f"tape {self.key!r}: step {self.t} allocates more activations than step 0"
                bufs.append(torch.empty(full, dtype=dtype, device=device))
            b = bufs[i]
            if tuple(b.shape) != full or b.dtype != dtype:
                if self.t != 0:
                    raise RuntimeError(f"tape {self.key!r}: allocation {i} of step {self.t} is {tuple(shape)} {dtype}, step 0 had "
                                       f"{(b.shape[0] // self.T,) + tuple(b.shape[1:])} {b.dtype} (the steps of a tape must have one shape)")
                _KEEPALIVE.append(b)                    # an older capture may still write here
                b = bufs[i] = torch.empty(full, dtype=dtype, device=device)
            return b[self.t * shape[0]:(self.t + 1) * shape[0]]
        n = self.steps.get(self.key, 0)
        assert i < len(bufs) and n > 0, f"tape {self.key!r}: the ghost pass allocates activation {i}, the steps recorded {len(bufs)}"
        b = bufs[i]
        per = b.shape[0] // self.T
        if tuple(shape) != (n * per,) + tuple(b.shape[1:]) or b.dtype != dtype:
            raise RuntimeError(f"tape {self.key!r}: ghost allocation {i} is {tuple(shape)} {dtype}, the {n} recorded steps hold "
                               f"{(n * per,) + tuple(b.shape[1:])} {b.dtype}")
        return b[:n * per]

    def seed(self, n):
        seeds, si = self.seeds[self.key], self.si
        self.si += 1
        if self.mode == "record" and self.t == 0:
            assert si == len(seeds)
            seeds.append(next_seeds(n))
        return seeds[si]


class _TapeUse:
    def __init__(self, tape, key):
        self.tape, self.key, self.prev = tape, key, None

    def __enter__(self):
        self.prev = self.tape._swap(self.key)
        return self.tape

    def __exit__(self, *exc):
        self.tape._swap(self.prev)
        return False


class _TapeCtx:
    def __init__(self, tape, keys, mode, t, batch_n=None):
        self.a = (tape, keys, mode, t)
        self.batch_n = batch_n
        self.ng = torch.no_grad() if mode == "record" else None

    def __enter__(self):
        tape, keys, mode, t = self.a
        tape._enter(keys, mode, t, self.batch_n)
        if self.ng is not None:
            self.ng.__enter__()
        return tape

    def __exit__(self, *exc):
        tape = self.a[0]
        if self.ng is not None:
            self.ng.__exit__(*exc)
        counts = tape._exit()
        if exc[0] is None and self.a[2] != "compute":
            for key, (n_alloc, n_seed) in counts.items():
                assert n_alloc == len(tape.bufs[key]) and n_seed == len(tape.seeds[key]), \
                    f"tape {key!r}: {self.a[2]} pass made {n_alloc} allocations / {n_seed} seed draws, step 0 made " \
                    f"{len(tape.bufs[key])} / {len(tape.seeds[key])}"
        return False


def _new(shape, dtype, device):
    """Activation buffer of a FORWARD operator: plain torch.empty, or the episode tape's slice / whole buffer."""
    if _TAPE is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return _TAPE.take(tuple(shape), dtype, device)


def _ghost():
    return _TAPE is not None and _TAPE.mode == "ghost"


def _shift(seed, step_elems):
    """Dropout seed of a step's launch inside a recording tape (see above); step_elems = elements of the masked tensor per step."""
    if _TAPE is None or _TAPE.mode != "record" or _TAPE.t == 0:
        return seed
    return (seed + _TAPE.t * step_elems * _HASH_MUL) & 0xFFFFFFFF


# =====================================================================================
#  raw kernel wrappers (no autograd)
# =====================================================================================
# Per-shape kernel choice: the GEMM pipelines (register-staged 4 blocks/CU, LDS-DMA 2-/3-stage with 4 or 8 waves on 128x128
# tiles, LDS-DMA large tiles 256x128 / 256x256 / 128x256) compute identical results; which is fastest depends on how the tile
# count fills 256 CUs and on how many operand bytes each CU pulls per output. The first call of a shape
# times each once with HIP events (a few hundred microseconds) and the winner is cached for the life of the process.
AUTOTUNE = True
GEMM_VARIANTS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14)
P8_VARIANT = 15          # 256 x 256 "8-phase" persistent kernel (v_mfma_16x16x32, one block per CU): tried for long launches only
P8_MIN_ROWS = int(os.environ.get("VLNI_P8_MIN_ROWS", "6144"))
P8H_VARIANT = 32         # (id 16) 256 x 128 tiles, three LDS buffers, four loader waves: step-long launches and narrow projections
P8H_MIN_ROWS = int(os.environ.get("VLNI_P8H_MIN_ROWS", "1024"))
_GEMM_BEST = {}


def export_tune():
    """The kernel choices of this process (GEMM pipeline per launch shape, weight-gradient variant x split) as plain JSON-able lists."""
    enc = lambda k: [str(x) if isinstance(x, torch.dtype) else (list(x) if isinstance(x, tuple) else x) for x in k]
    return {"gemm": [[enc(k), v] for k, v in _GEMM_BEST.items()], "tn": [[list(k), list(v)] for k, v in _TN_BEST.items()],
            "tnb": [[list(k), list(v)] for k, v in _TNB_BEST.items()]}


def import_tune(obj):
    """Adopts choices written by export_tune() (bench.py --load-tune: the counter passes then launch the timed run's kernels)."""
    dts = {str(d): d for d in (torch.float32, torch.bfloat16, torch.float16)}
    dec = lambda k: tuple(dts.get(x, x) if isinstance(x, str) else (tuple(x) if isinstance(x, list) else x) for x in k)
    _GEMM_BEST.update({dec(k): v for k, v in obj["gemm"]})
    _TN_BEST.update({tuple(k): tuple(v) for k, v in obj["tn"]})
    _TNB_BEST.update({tuple(k): tuple(v) for k, v in obj["tnb"]})


def _nt_variants(rows, K, dtype):
    v = GEMM_VARIANTS
    if dtype in H16 and len(v) > 1:
        if rows >= P8H_MIN_ROWS and K % 64 == 0 and K >= 128 and P8H_VARIANT not in v:
            v = v + (P8H_VARIANT,)
        if rows >= P8_MIN_ROWS and K % 128 == 0 and P8_VARIANT not in v:
            v = v + (P8_VARIANT,)
    return v


def _pick(variants, launch):
    """Fastest of the candidate pipelines for one launch: every candidate once warm + 3 timed launches, then the three best again with 8
    launches each (a single 3-launch sample put the runner-up first on about one shape in five, and the step time moved by +-1 ms from
    run to run with it)."""
    def timed(v, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch(v)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n
    first = []
    for v in variants:
        launch(v)                                                   # warm
        first.append((timed(v, 4), v))
    first.sort()
    # (round 4: four finalists, 16 launches each, two rounds interleaved - the picks of two runs still differed on a third of the step's shapes
    # with three finalists at 8 launches, and the step moved by 0.5 ms with them)
    best = {}
    for _ in range(2):
        for _, v in first[:4]:
            best[v] = min(best.get(v, float("inf")), timed(v, 8))
    return min(best, key=best.get)


class KN:
    """Marks a weight handed to gemm_nt / gemm_nt2 as [K, N] row-major (the forward weight itself as the dgrad operand):
    the launch uses the transposing-read "NN" kernels (variant + 16) instead of a transposed weight copy."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


NN_DGRAD = os.environ.get("VLNI_NN_DGRAD", "0") == "1"    # (round 3: off. The 256-wide NT kernels on the W^T copy beat the transposing-read kernels on
                         # every launch of >= 1 k rows: 885-909 vs 670-690 TF/s at K = 2304 / 3072 on a step's rows, tools/gemm_step_probe.py; the
                         # copies cost one batched transpose per optimizer step.) bf16 dgrad straight from W[out, in] through the transposing-read kernel: no W^T
                         # shadows to rebuild after every optimizer step and half the shadow memory. Same-box A/B on the bench step
                         # (3 pairs): 37.5-37.8 ms vs 38.1 ms with NT kernels on W^T copies. (Before the LDS-DMA of that kernel was
                         # issued as asm it was the slower choice: the compiler serialised its copies with the reads.)
NN_VARIANTS = (2, 3, 4, 5, 6)
NN_MIN_ROWS = int(os.environ.get("VLNI_NN_MIN_ROWS", "4096"))   # below this the small-tile NT pipelines on a W^T copy win (DUET's map / viewpoint
                                                                # streams: 25.0 vs 26.0 ms per step with every dgrad on the NN kernel)


NT_LONG_ROWS = int(os.environ.get("VLNI_NT_LONG_ROWS", "16384"))   # episode-batched dgrad launches (30-50 k rows): the 256 x 256 8-phase NT kernel
                                                                   # on the W^T copy beats the transposing-read kernels there


class WT:
    """The dgrad operand of a (possibly row-packed) weight, resolved by gemm_nt / gemm_nt2 once the row count of the launch is known:
    the weight itself as [K, N] (KN, transposing-read kernel) for step-long launches, the maintained W^T copy for short ones and for
    episode-long ones (NT_LONG_ROWS)."""
    __slots__ = ("params", "dtype")

    def __init__(self, params, dtype):
        self.params, self.dtype = params, dtype

    @property
    def n(self):                      # output width of the dgrad launch = in_features
        return self.params[0].shape[1]

    def resolve(self, rows):
        if NN_MIN_ROWS <= rows < NT_LONG_ROWS:
            return KN(SHADOWS.get(self.params, self.dtype, False))
        return SHADOWS.get(self.params, self.dtype, True)


def _gemm_call(variant, a, b, out, bias, act, residual, preact, dact_src, dact, alpha, split_k, atomic, M, N, K, drop=None):
    _lib.call("vlni_gemm_nt_v", _dt(a), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
              out.stride(0), M, N, K, _p(bias), act, _p(residual), residual.stride(0) if residual is not None else 0,
              _p(preact), preact.stride(0) if preact is not None else 0, _p(dact_src),
              dact_src.stride(0) if dact_src is not None else 0, dact, alpha, split_k, 1 if atomic else 0, variant,
              drop[0] if drop else 0.0, drop[1] if drop else 0, _st())


def gemm_nt(a, b, out=None, bias=None, act=0, residual=None, preact=None, dact_src=None, dact=0,
            alpha=1.0, split_k=1, atomic=False, out_dtype=None, drop=None):
    """out[M,N] = epi(a[M,K] @ b[N,K]^T); a, b same dtype, K-contiguous. drop = (p, seed): dropout after act, before residual."""
    M, K = a.shape
    if isinstance(b, WT):
        b = b.resolve(M)
    kn = isinstance(b, KN)
    if kn:
        b = b.t
        N = b.shape[1]
        assert b.shape[0] == K and a.dtype == b.dtype and a.dtype in H16 and a.stride(1) == 1 and b.stride(1) == 1 and not atomic
    else:
        N = b.shape[0]
        assert b.shape[1] == K and a.dtype == b.dtype and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = _new((M, N), torch.float32 if atomic else a.dtype, a.device)
    if _ghost():
        return out
    if drop is not None and drop[0] <= 0.0:
        drop = None
    if drop is not None:
        drop = (drop[0], _shift(drop[1], M * N))
    args = (a, b, out, bias, act, residual, preact, dact_src, dact, alpha, split_k, atomic, M, N, K, drop)
    variant = 21 if kn else 0
    if AUTOTUNE and not atomic and M >= 512:
        key = (a.dtype, M, N, K, act, dact, residual is not None, preact is not None, kn)
        variant = _GEMM_BEST.get(key)
        if variant is None and torch.cuda.is_current_stream_capturing():
            variant = 21 if kn else 0                                       # no timing trials inside a graph capture
        elif variant is None:
            variant = _GEMM_BEST[key] = _pick([16 + u for u in NN_VARIANTS] if kn else _nt_variants(M, K, a.dtype),
                                              lambda v: _gemm_call(v, *args))
    _gemm_call(variant, *args)
    return out


def _arr(ctype, vals):
    return (ctype * 2)(*vals)


GELU_STORE_GRAD = os.environ.get("VLNI_GELU_STORE_GRAD", "1") == "1"


def _gelu_codes(dtype):
    """(act, dact) of an FFN's two fused epilogues. 16-bit paths: the forward stores GELU'(z) where it would store z (act 3) and the dgrad epilogue
    multiplies by that tensor (dact 3) - GELU' cost the episode-long FFN dgrad launches 40 % (three transcendentals per element in a 256 x 256
    tile's epilogue that no other block hides). float32 (the parity path) keeps z and erf."""
    return (3, 3) if (GELU_STORE_GRAD and dtype in H16) else (1, 1)


def gemm_nt2(a, b, bias=(None, None), act=0, residual=(None, None), preact=(None, None), dact_src=(None, None), dact=0,
             drop=None):
    """Two GEMMs with the same (N, K) and epilogue kind in ONE launch: out_i = epi(a_i @ b_i^T), i = 0, 1.
    drop = (p, (seed0, seed1)). Returns (out0, out1)."""
    (a0, a1), (b0, b1) = a, b
    M0, K = a0.shape
    M1 = a1.shape[0]
    if isinstance(b0, WT):
        b0, b1 = b0.resolve(M0 + M1), b1.resolve(M0 + M1)
        b = (b0, b1)
    kn = isinstance(b0, KN)
    if kn:
        b0, b1 = b0.t, b1.t
        b = (b0, b1)
        N = b0.shape[1]
        assert b0.shape[0] == K and a0.dtype in H16
    else:
        N = b0.shape[0]
    assert b1.shape == b0.shape and a1.shape[1] == K and a0.dtype == a1.dtype == b0.dtype == b1.dtype
    outs = (_new((M0, N), a0.dtype, a0.device), _new((M1, N), a0.dtype, a0.device))
    if _ghost():
        return outs
    vp, lg = ctypes.c_void_p, ctypes.c_long
    ptr2 = lambda ts: _arr(vp, [_p(t) for t in ts])
    ld2 = lambda ts: _arr(lg, [t.stride(0) if t is not None else 0 for t in ts])
    if drop is not None and drop[0] <= 0.0:
        drop = None
    seeds = _arr(ctypes.c_uint, (_shift(drop[1][0], M0 * N), _shift(drop[1][1], M1 * N)) if drop else (0, 0))
    cargs = (_dt(a0), ptr2(a), ld2(a), ptr2(b), ld2(b), ptr2(outs), ld2(outs), _arr(ctypes.c_int, (M0, M1)), N, K,
             ptr2(bias), act, ptr2(residual), ld2(residual), ptr2(preact), ld2(preact), ptr2(dact_src), ld2(dact_src), dact)
    launch = lambda v: _lib.call("vlni_gemm_nt_dual", *cargs, v, drop[0] if drop else 0.0, seeds, _st())
    variant = 21 if kn else 0
    if AUTOTUNE:
        key = (a0.dtype, M0, M1, N, K, act, dact, residual[0] is not None, preact[0] is not None, kn)
        variant = _GEMM_BEST.get(key)
        if variant is None and torch.cuda.is_current_stream_capturing():
            variant = 21 if kn else 0
        elif variant is None:
            variant = _GEMM_BEST[key] = _pick([16 + u for u in NN_VARIANTS] if kn else _nt_variants(M0 + M1, K, a0.dtype), launch)
    launch(variant)
    return outs


def _use(key):
    """Allocations / seed draws of one problem of a lockstep recorded step go to tape key `key` (None: the active key)."""
    import contextlib
    return _TAPE.use(key) if (_TAPE is not None and key is not None) else contextlib.nullcontext()


class fork:
    """`with fork(side): ...` queues the body on stream `side` behind everything queued on the current stream so far; join() makes the current
    stream wait for it. Inside a captured step these are graph edges: two small kernels of independent problems (the panorama's attention
    beside the two streams' dual attention, its LayerNorm beside theirs) run side by side instead of back to back. side=None: no-op."""

    def __init__(self, side):
        self.side = side
        self.main = torch.cuda.current_stream() if side is not None else None
        self.ctx = None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def gemm_ntn(a, b, bias=None, act=0, residual=None, preact=None, dact_src=None, dact=0, drop=None, keys=None):
    """n <= 4 GEMMs with the same (N, K) and epilogue kind in ONE launch (vlni_gemm_nt_multi): out_i = epi(a_i @ b_i^T). Every per-problem
    argument is a sequence of n; drop = (p, seeds); keys: per problem the episode-tape key its output belongs to (lockstep recorded
    steps, EpisodeTape.use). Returns the n outputs. n = 2 is gemm_nt2's launch."""
    n = len(a)
    none = (None,) * n
    bias, residual, preact, dact_src, keys = bias or none, residual or none, preact or none, dact_src or none, keys or none
    K = a[0].shape[1]
    N = b[0].shape[0]
    dt = a[0].dtype
    assert 1 <= n <= 4 and all(x.shape[1] == K and x.dtype == dt and x.stride(1) == 1 for x in a)
    assert all(w.shape == b[0].shape and w.dtype == dt and w.stride(1) == 1 for w in b) and b[0].shape[1] == K
    outs = []
    for i in range(n):
        with _use(keys[i]):
            outs.append(_new((a[i].shape[0], N), dt, a[i].device))
    if _ghost():
        return tuple(outs)
    vp, lg = ctypes.c_void_p, ctypes.c_long
    ptrs = lambda ts: (vp * n)(*[_p(t) for t in ts])
    lds = lambda ts: (lg * n)(*[t.stride(0) if t is not None else 0 for t in ts])
    Ms = [x.shape[0] for x in a]
    if drop is not None and drop[0] <= 0.0:
        drop = None
    seeds = (ctypes.c_uint * n)(*([_shift(drop[1][i], Ms[i] * N) for i in range(n)] if drop else [0] * n))
    cargs = (_dt(a[0]), n, ptrs(a), lds(a), ptrs(b), lds(b), ptrs(outs), lds(outs), (ctypes.c_int * n)(*Ms), N, K,
             ptrs(bias), act, ptrs(residual), lds(residual), ptrs(preact), lds(preact), ptrs(dact_src), lds(dact_src), dact)
    launch = lambda v: _lib.call("vlni_gemm_nt_multi", *cargs, v, drop[0] if drop else 0.0, seeds, _st())
    variant = 0
    if AUTOTUNE:
        key = (dt, tuple(Ms), N, K, act, dact, residual[0] is not None, preact[0] is not None, "n")
        variant = _GEMM_BEST.get(key)
        if variant is None and torch.cuda.is_current_stream_capturing():
            variant = 0
        elif variant is None:
            variant = _GEMM_BEST[key] = _pick(_nt_variants(sum(Ms), K, dt), launch)
    launch(variant)
    return tuple(outs)


def rec_self_att3(xs, kms, drops, Ps, eps, keys, side=None):
    """RECORD-mode forward (episode tape, no autograd) of _DualSelfAttBlock(x0, x1) + _SelfAttBlock(x2) in lockstep: the two projections of all
    three problems run as 3-problem GEMM launches (the history panorama encoder's layer beside the language / vision streams of the
    cross-modal layer with the same (N, K, epilogue): vilmodel_cmt.py:399-407 and :216-239 under :603-614). keys = (key of problems 0 / 1,
    key of problem 2): every activation is allocated under its own call's tape key, in the order that call's ghost pass asks for it
    (_DualSelfAttBlock.forward / _SelfAttBlock.forward), so the episode-batched backward runs the UNMERGED autograd nodes on these buffers."""
    assert _TAPE is not None and _TAPE.mode == "record"
    kv, kh = keys
    k3 = (kv, kv, kh)
    x0, x1, x2 = xs
    (B, S0, H), S1, (B2, S2, _) = x0.shape, x1.shape[1], x2.shape
    a = [_rows(_chk(x, "x")) for x in xs]
    dt = x0.dtype
    wqkv = [_w((P[0], P[2], P[4]), dt) for P in Ps]
    bqkv = [_w((P[1], P[3], P[5]), torch.float32) for P in Ps]
    q = gemm_ntn(a, wqkv, bias=bqkv, keys=k3)
    d0, d1, d2 = drops
    f = fork(side)                         # problem 2's attention (and LayerNorm below) beside the dual launch of problems 0 / 1
    with f, _use(kh):
        c2, _ = attn_fwd(q[2][:, :H], q[2][:, H:2 * H], q[2][:, 2 * H:], B2, S2, S2, kms[2], None, drop=(d2[0], d2[2]))
    with _use(kv):
        (c0, _), (c1, _) = attn_fwd2((q[0][:, :H], q[1][:, :H]), (q[0][:, H:2 * H], q[1][:, H:2 * H]), (q[0][:, 2 * H:], q[1][:, 2 * H:]), B,
                                     (S0, S1), (S0, S1), (kms[0], kms[1]), None, drop=(max(d0[0], d1[0]), (d0[2], d1[2])))
    f.join()
    ph = max(d0[1], d1[1])
    assert ph == d2[1], "lockstep blocks: one hidden-dropout probability"
    pre = gemm_ntn((c0, c1, c2), [_w((P[6],), dt) for P in Ps], bias=[P[7] for P in Ps], residual=a,
                   drop=(ph, (d0[2] + 1, d1[2] + 1, d2[2] + 1)), keys=k3)
    f = fork(side)
    with f, _use(kh):
        y2, _, _ = ln_fwd(pre[2], Ps[2][8], Ps[2][9], eps)
    with _use(kv):
        (y0, _, _), (y1, _, _) = ln_fwd2((pre[0], pre[1]), (Ps[0][8], Ps[1][8]), (Ps[0][9], Ps[1][9]), eps)
    f.join()
    return y0.view(B, S0, H), y1.view(B, S1, H), y2.view(B2, S2, H)


def rec_ffn3(xs, drops, Ps, eps, keys, side=None):
    """RECORD-mode forward of _DualFfnBlock(x0, x1) + _FfnBlock(x2) in lockstep (see rec_self_att3): FFN-in and FFN-out as 3-problem launches."""
    assert _TAPE is not None and _TAPE.mode == "record"
    kv, kh = keys
    k3 = (kv, kv, kh)
    shp = [x.shape for x in xs]
    a = [_rows(_chk(x, "x")) for x in xs]
    dt = xs[0].dtype
    FF = Ps[0][0].shape[0]
    z = []
    for i in range(3):
        with _use(k3[i]):
            z.append(_new((a[i].shape[0], FF), dt, a[i].device))
    h = gemm_ntn(a, [_w((P[0],), dt) for P in Ps], bias=[P[1] for P in Ps], act=_gelu_codes(dt)[0], preact=z, keys=k3)
    d0, d1, d2 = drops
    ph = max(d0[1], d1[1])
    assert ph == d2[1], "lockstep blocks: one hidden-dropout probability"
    pre = gemm_ntn(h, [_w((P[2],), dt) for P in Ps], bias=[P[3] for P in Ps], residual=a, drop=(ph, (d0[2], d1[2], d2[2])), keys=k3)
    f = fork(side)
    with f, _use(kh):
        y2, _, _ = ln_fwd(pre[2], Ps[2][4], Ps[2][5], eps)
    with _use(kv):
        (y0, _, _), (y1, _, _) = ln_fwd2((pre[0], pre[1]), (Ps[0][4], Ps[1][4]), (Ps[0][5], Ps[1][5]), eps)
    f.join()
    return y0.view(shp[0]), y1.view(shp[1]), y2.view(shp[2])


def transpose_pad(x, rpad, dtype=None):
    """[R,C] -> [C,rpad] (zero-padded columns R..rpad), optional dtype change."""
    R, C = x.shape
    dtype = dtype or x.dtype
    out = torch.empty((C, rpad), dtype=dtype, device=x.device)
    _lib.call("vlni_transpose", _dt(x), _DT[dtype], x.data_ptr(), x.stride(0), out.data_ptr(), rpad, R, C, rpad, _st())
    return out


def wgrad(dy, x, out=None, colsum_out=None, want_colsum=False):
    """dW[N,K] (+)= dy[M,N]^T @ x[M,K] in float32 (atomics, split over rows); optionally also the bias gradient
    db[N] (+)= column sums of dy. bf16: transposing-read TN kernel straight on dy / x (colsum fused);
    fp32 (parity path): explicit transposes + the NT kernel + a colsum pass. Returns dW or (dW, db)."""
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=dev)
    if (want_colsum or colsum_out is not None) and colsum_out is None:
        colsum_out = torch.zeros((N,), dtype=torch.float32, device=dev)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    if dy.dtype in H16 and N % 8 == 0 and K % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0:
        split = max(1, min(8, ((M + 63) // 64) // 8, round(768 / tiles)))
        _lib.call("vlni_gemm_tn_h16_grouped_v", _dt(dy), 1, (ctypes.c_void_p * 1)(dy.data_ptr()), (ctypes.c_void_p * 1)(x.data_ptr()),
                  (ctypes.c_int * 1)(M), dy.stride(0), x.stride(0), out.data_ptr(), out.stride(0), N, K, _p(colsum_out), split, 0, _st())
    else:
        mp = (M + 63) // 64 * 64
        dyt = transpose_pad(dy, mp)
        xt = transpose_pad(x, mp)
        bk = 64 if dy.dtype in H16 else 32
        nkt = (mp + bk - 1) // bk
        split = max(1, min(nkt, 8, round(512 / tiles)))
        gemm_nt(dyt, xt, out=out, split_k=split, atomic=True)
        if colsum_out is not None:
            colsum(dy, out=colsum_out)
    return (out, colsum_out) if colsum_out is not None else out


def colsum(x, out=None):
    rows, N = x.shape
    if out is None:
        out = torch.zeros((N,), dtype=torch.float32, device=x.device)
    _lib.call("vlni_colsum", _dt(x), x.data_ptr(), x.stride(0), rows, N, out.data_ptr(), _st())
    return out


def ln_fwd(x, gamma, beta, eps):
    rows, H = x.shape
    y = _new((rows, H), x.dtype, x.device)
    mean = _new((rows,), torch.float32, x.device)
    rstd = _new((rows,), torch.float32, x.device)
    if _ghost():
        return y, mean, rstd
    _lib.call("vlni_layernorm_fwd", _dt(x), x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), eps,
              y.data_ptr(), y.stride(0), mean.data_ptr(), rstd.data_ptr(), rows, H, _st())
    return y, mean, rstd


def ln_bwd(dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, want_param_grads=True, dres=None, drop=None):
    """drop = (p, seed): additionally returns dx * mask/(1-p) as 4th value (grad of the dropped dense output)."""
    rows, H = x.shape
    dx = torch.empty((rows, H), dtype=x.dtype, device=x.device)
    dxd = torch.empty((rows, H), dtype=x.dtype, device=x.device) if (drop is not None and drop[0] > 0.0) else None
    if want_param_grads and dgamma is None:
        dgamma = torch.zeros((H,), dtype=torch.float32, device=x.device)
        dbeta = torch.zeros((H,), dtype=torch.float32, device=x.device)
    _lib.call("vlni_layernorm_bwd", _dt(x), dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(),
              mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dx.stride(0), _p(dgamma), _p(dbeta), rows, H,
              _p(dres), dres.stride(0) if dres is not None else 0, _p(dxd), H if dxd is not None else 0,
              drop[0] if dxd is not None else 0.0, drop[1] if dxd is not None else 0, _st())
    if drop is not None:
        return dx, dgamma, dbeta, (dxd if dxd is not None else dx)
    return dx, dgamma, dbeta


def _i2(a, b):
    return (ctypes.c_int * 2)(a, b)


def ln_fwd2(xs, gs, bs, eps):
    """Two LayerNorms (tuples of 2: the two streams of a cross-modal layer) in ONE launch. Returns ((y0, mean0, rstd0), (y1, mean1, rstd1))."""
    H = xs[0].shape[1]
    if xs[0].dtype != xs[1].dtype or xs[1].shape[1] != H:
        return ln_fwd(xs[0], gs[0], bs[0], eps), ln_fwd(xs[1], gs[1], bs[1], eps)
    ys = tuple(_new((x.shape[0], H), x.dtype, x.device) for x in xs)
    ms = tuple(_new((x.shape[0],), torch.float32, x.device) for x in xs)
    rs = tuple(_new((x.shape[0],), torch.float32, x.device) for x in xs)
    if _ghost():
        return (ys[0], ms[0], rs[0]), (ys[1], ms[1], rs[1])
    _lib.call("vlni_layernorm_fwd_dual", _dt(xs[0]), _p2(xs), _l2(xs), _p2(gs), _p2(bs), eps, _p2(ys), _l2(ys), _p2(ms), _p2(rs),
              _i2(xs[0].shape[0], xs[1].shape[0]), H, _st())
    return (ys[0], ms[0], rs[0]), (ys[1], ms[1], rs[1])


def _ln_bwd_to2(dys, xs, gs, bs, means, rstds, wants, drop=None):
    """Two LayerNorm backwards in ONE launch; per problem the return value of _ln_bwd_to: (dx, dgamma, dbeta[, dx_dropped]) with
    dgamma / dbeta None where they were accumulated in place. drop = (p, (seed0, seed1))."""
    H = xs[0].shape[1]
    if xs[0].dtype != xs[1].dtype or xs[1].shape[1] != H:
        return tuple(_ln_bwd_to(dys[i], xs[i], gs[i], bs[i], means[i], rstds[i], wants[i],
                                drop=(drop[0], drop[1][i]) if drop is not None else None) for i in range(2))
    dev = xs[0].device
    dxs = tuple(torch.empty((x.shape[0], H), dtype=x.dtype, device=dev) for x in xs)
    dropping = drop is not None and drop[0] > 0.0
    dxd = tuple(torch.empty((x.shape[0], H), dtype=x.dtype, device=dev) for x in xs) if dropping else (None, None)
    dgs, dbs, ret = [], [], []
    for i in range(2):
        if wants[i] and _direct(gs[i], bs[i]):
            dgs.append(gs[i].grad); dbs.append(bs[i].grad); ret.append(False)
        elif wants[i]:
            dgs.append(torch.zeros((H,), dtype=torch.float32, device=dev)); dbs.append(torch.zeros((H,), dtype=torch.float32, device=dev)); ret.append(True)
        else:
            dgs.append(None); dbs.append(None); ret.append(False)
    _lib.call("vlni_layernorm_bwd_dual", _dt(xs[0]), _p2(dys), _l2(dys), _p2(xs), _l2(xs), _p2(gs), _p2(means), _p2(rstds), _p2(dxs), _l2(dxs),
              _p2(dgs), _p2(dbs), _i2(xs[0].shape[0], xs[1].shape[0]), H, None, None, _p2(dxd) if dropping else None,
              (ctypes.c_long * 2)(H, H) if dropping else None, drop[0] if dropping else 0.0,
              (ctypes.c_uint * 2)(*drop[1]) if dropping else None, _st())
    out = []
    for i in range(2):
        r = (dxs[i], dgs[i] if ret[i] else None, dbs[i] if ret[i] else None)
        if drop is not None:
            r = r + (dxd[i] if dropping else dxs[i],)
        out.append(r)
    return tuple(out)


def attn_fwd(q, k, v, B, Sq, Sk, kmask=None, bias=None, nh=12, drop=None):
    """q [B*Sq, >=nh*64] (strided view), k/v [B*Sk, ...]; returns ctx [B*Sq, nh*64], lse [B,nh,Sq]."""
    out = _new((B * Sq, nh * 64), q.dtype, q.device)
    lse = _new((B, nh, Sq), torch.float32, q.device)
    if _ghost():
        return out, lse
    _lib.call("vlni_attn_fwd", _dt(q), q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
              _p(kmask), _p(bias), out.data_ptr(), out.stride(0), lse.data_ptr(), B, nh, Sq, Sk, 1.0 / 8.0,
              drop[0] if drop else 0.0, _shift(drop[1], B * nh * Sq * Sk) if drop else 0, _st())
    return out, lse


def attn_probs(q, k, B, Sq, Sk, kmask=None, bias=None, nh=12):
    """softmax(q k^T / 8 + kmask (+ bias)) as float32 [B, nh, Sq, Sk] - visualisation only (vlni_attn_probs)."""
    P = torch.empty((B, nh, Sq, Sk), dtype=torch.float32, device=q.device)
    _lib.call("vlni_attn_probs", _dt(q), q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), _p(kmask), _p(bias), P.data_ptr(),
              B, nh, Sq, Sk, 1.0 / 8.0, _st())
    return P


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, Sq, Sk, kmask=None, bias=None, dbias=None, nh=12, drop=None):
    _lib.call("vlni_attn_bwd", _dt(q), q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
              _p(kmask), _p(bias), out.data_ptr(), out.stride(0), dout.data_ptr(), dout.stride(0), lse.data_ptr(),
              dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), _p(dbias),
              B, nh, Sq, Sk, 1.0 / 8.0, drop[0] if drop else 0.0, drop[1] if drop else 0, _st())


def _dual_attn_ok(*ts):
    """bf16 operands whose rows start on 16-byte boundaries with strides that are multiples of 8: the dual-problem launch."""
    return all(t.dtype in H16 and t.dtype == ts[0].dtype and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 and t.stride(1) == 1 for t in ts)


def _p2(ts):
    return (ctypes.c_void_p * 2)(*[_p(t) for t in ts])


def _l2(ts):
    return (ctypes.c_long * 2)(*[t.stride(0) if t is not None else 0 for t in ts])


def attn_fwd2(q, k, v, B, Sq, Sk, kmask=(None, None), bias0=None, nh=12, drop=None):
    """Two attention problems (tuples of 2: the two streams / directions of a cross-modal layer) in ONE launch (vlni_attn_fwd_dual);
    drop = (p, (seed0, seed1)). Returns ((ctx0, lse0), (ctx1, lse1)). Falls back to two launches outside the bf16 fast path."""
    p_, seeds = (drop[0], drop[1]) if drop else (0.0, (0, 0))       # one dropout probability for both (it is one config value)
    if not _dual_attn_ok(*q, *k, *v) or max(Sk) > 256:
        return (attn_fwd(q[0], k[0], v[0], B, Sq[0], Sk[0], kmask[0], bias0, nh, drop=(p_, seeds[0])),
                attn_fwd(q[1], k[1], v[1], B, Sq[1], Sk[1], kmask[1], None, nh, drop=(p_, seeds[1])))
    outs = tuple(_new((B * Sq[i], nh * 64), q[i].dtype, q[i].device) for i in range(2))
    lses = tuple(_new((B, nh, Sq[i]), torch.float32, q[i].device) for i in range(2))
    if _ghost():
        return (outs[0], lses[0]), (outs[1], lses[1])
    seeds = tuple(_shift(seeds[i], B * nh * Sq[i] * Sk[i]) for i in range(2))
    _lib.call("vlni_attn_fwd_dual", _dt(q[0]), _p2(q), _l2(q), _p2(k), _l2(k), _p2(v), _l2(v), _p2(kmask), _p2((bias0, None)), _p2(outs), _l2(outs),
              _p2(lses), B, nh, (ctypes.c_int * 2)(*Sq), (ctypes.c_int * 2)(*Sk), 1.0 / 8.0, p_, (ctypes.c_uint * 2)(*seeds), _st())
    return (outs[0], lses[0]), (outs[1], lses[1])


def attn_bwd2(q, k, v, out, dout, lse, dq, dk, dv, B, Sq, Sk, kmask=(None, None), bias0=None, dbias0=None, nh=12, drop=None):
    """Backward of attn_fwd2 in one launch (vlni_attn_bwd_dual); every argument but bias0 / dbias0 is a tuple of 2."""
    p_, seeds = (drop[0], drop[1]) if drop else (0.0, (0, 0))
    if not _dual_attn_ok(*q, *k, *v, *dout) or max(Sk) > 256:
        attn_bwd(q[0], k[0], v[0], out[0], dout[0], lse[0], dq[0], dk[0], dv[0], B, Sq[0], Sk[0], kmask[0], bias0, dbias0, nh, drop=(p_, seeds[0]))
        attn_bwd(q[1], k[1], v[1], out[1], dout[1], lse[1], dq[1], dk[1], dv[1], B, Sq[1], Sk[1], kmask[1], None, None, nh, drop=(p_, seeds[1]))
        return
    _lib.call("vlni_attn_bwd_dual", _dt(q[0]), _p2(q), _l2(q), _p2(k), _l2(k), _p2(v), _l2(v), _p2(kmask), _p2((bias0, None)), _p2(out), _l2(out),
              _p2(dout), _l2(dout), _p2(lse), _p2(dq), _l2(dq), _p2(dk), _l2(dk), _p2(dv), _l2(dv), _p(dbias0), B, nh,
              (ctypes.c_int * 2)(*Sq), (ctypes.c_int * 2)(*Sk), 1.0 / 8.0, p_, (ctypes.c_uint * 2)(*seeds), _st())


def dropout_apply(x, p, seed):
    """x * mask/(1-p) with the library's counter-based mask over the linear index (x contiguous)."""
    if p <= 0.0:
        return x
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.call("vlni_dropout", _dt(x), x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _st())
    return y


_SEED = [None]


def next_seeds(n=4):
    """n fresh 31-bit dropout seeds (deterministic after torch.manual_seed)."""
    if _SEED[0] is None:
        _SEED[0] = torch.initial_seed() & 0x3FFFFFFF
    base = _SEED[0]
    _SEED[0] = (base + n * 0x9E37 + 1) & 0x3FFFFFFF
    return base


def reseed(seed):
    _SEED[0] = seed & 0x3FFFFFFF


_SEED_BASE = [None]


def set_seed_base(t):
    """Registers a 1-element int32 CUDA tensor whose value every kernel adds to its dropout seed (None = off). A captured
    training step advances it in-graph (train.GraphedStep), so each replay draws fresh masks."""
    assert t is None or (t.is_cuda and t.dtype == torch.int32 and t.numel() == 1)
    _lib.call("vlni_set_dropout_seed_base", 0 if t is None else t.data_ptr())
    _SEED_BASE[0] = t


_INDEX_ERRORS = [None]


def watch_index_errors(device=None):
    """Registers (once per process) the device counter of out-of-range table indices. The gather / scatter kernels that take caller-supplied
    row indices (navigation types, step ids, positions: vlni_embed_combine_fwd, vlni_scatter_add_rows*) skip such a row instead of reading
    or adding out of bounds, and count it here - nn.Embedding, which they replace, raised."""
    if _INDEX_ERRORS[0] is None:
        _INDEX_ERRORS[0] = torch.zeros(1, dtype=torch.int32, device=device or "cuda")
        _lib.call("vlni_set_index_error_counter", _INDEX_ERRORS[0].data_ptr())
    return _INDEX_ERRORS[0]


def index_errors(reset=True, raise_=True):
    """Rows skipped because their table index was out of range since the last call (one device-to-host read: call it where a sync is
    acceptable - end of an iteration, a test). Raises IndexError like the reference's nn.Embedding lookups unless raise_=False."""
    t = _INDEX_ERRORS[0]
    if t is None:
        return 0
    n = int(t.item())
    if n and reset:
        t.zero_()
    if n and raise_:
        raise IndexError(f"{n} embedding row index(es) out of range (navigation-type / step / position ids): those rows contributed nothing")
    return n


def cast(x, dtype, tape=False):
    """tape=True: an ACTIVATION cast inside a forward operator (episode-tape aware); weights and gradients are never taped."""
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = _new(x.shape, dtype, x.device) if tape else torch.empty(x.shape, dtype=dtype, device=x.device)
    if tape and _ghost():
        return out
    _lib.call("vlni_cast", _dt(x), _DT[dtype], x.data_ptr(), out.data_ptr(), x.numel(), _st())
    return out


def additive_mask(m, inf=False):
    """bool/0-1 mask [B,S] -> float32 additive (1-m)*-10000 (vilmodel_cmt.py:1010-1012). One kernel for bool masks.
    inf=True: -inf instead of -10000 (nn.MultiheadAttention's key_padding_mask, VLN-DUET/map_nav_src/models/transformer.py:71-89)."""
    if m.dtype == torch.bool:
        return torch.where(m, _ZERO_NEG[0], _ZERO_NEG[2 if inf else 1])
    if inf:
        return torch.where(m != 0, _ZERO_NEG[0], _ZERO_NEG[2])
    return (1.0 - m.to(torch.float32)) * NEG_MASK


class _Consts:
    """float32 scalar tensors per device for torch.where (python scalars would make it a float64 -> float32 round trip)."""

    def __init__(self):
        self._c = {}

    def __getitem__(self, i):
        dev = torch.cuda.current_device()
        c = self._c.get(dev)
        if c is None:
            c = self._c[dev] = (torch.zeros((), dtype=torch.float32, device="cuda"), torch.full((), NEG_MASK, dtype=torch.float32, device="cuda"),
                                torch.full((), float("-inf"), dtype=torch.float32, device="cuda"))
        return c[i]


_ZERO_NEG = _Consts()


# =====================================================================================
#  compute-dtype shadows of the float32 master weights
# =====================================================================================
def _packed_param(params):
    """One [sum rows, ...] view over parameters that sit back to back in memory (train.FlatTrainer's arena), else None."""
    p0 = params[0]
    ptr, es, base = p0.data_ptr(), p0.element_size(), p0.untyped_storage().data_ptr()
    for p in params:
        if p.data_ptr() != ptr or not p.is_contiguous() or p.untyped_storage().data_ptr() != base:
            return None                # neighbours by accident (separate allocations) do not count
        ptr += p.numel() * es
    rows = sum(p.shape[0] for p in params)
    d = p0.detach()
    return torch.as_strided(d, (rows,) + tuple(d.shape[1:]), d.stride(), d.storage_offset())


class ShadowCache:
    """Packed / cast / transposed copies of parameters. The float32 nn.Parameters stay the single source of truth so
    state_dict keys match the reference checkpoints.

    Two staleness counters: `epoch` moves when parameters changed in a way nobody mirrored (load_state_dict, a test writing the
    arena: call invalidate()), `opt_epoch` moves after every optimizer step. With train.FlatTrainer's arenas registered
    (set_arena) the optimizer kernel itself keeps a bfloat16 mirror of the float32 arena, parameters that are neighbours in the
    arena are packed by a VIEW, so after a step only the transposed copies (dgrad operands) are rebuilt - from the bf16 mirror."""

    def __init__(self):
        self._c = {}
        self.epoch = 0
        self.opt_epoch = 0
        self.arena = None              # (flat float32 params, flat bf16 mirror)
        self._tr = {}                  # key -> (mirror view, transposed copy): refreshed together, one launch per optimizer step
        self._tr_table = None          # device table of vlni_transpose_batched (+ n, total tiles); None = rebuild needed
        self._tr_table_old = None      # the last table handed to a launch (kept alive when it is replaced)
        self._plain_tables = {}        # (dtype, (parameter, copy) addresses) -> device table of vlni_shadow_refresh

    def set_arena(self, flat_p, flat_b):
        self.arena = (flat_p, flat_b) if flat_p is not None else None
        self.epoch += 1
        self._tr, self._tr_table = {}, None

    def prepare(self):
        """Builds the device table of the batched transposed-shadow refresh (a host->device copy, so not inside a graph capture)."""
        live = {k: v for k, v in self._tr.items() if k in self._c and self._c[k][4]() is not None and self._c[k][1] is v[1]}
        if len(live) != len(self._tr):
            self._tr, self._tr_table = live, None
        if self._tr_table is not None or not self._tr or torch.cuda.is_current_stream_capturing():
            return self._tr_table is not None
        import struct
        buf, tile0 = bytearray(), 0
        for src, dst in self._tr.values():
            R, C = src.shape
            tiles_c = (C + 63) // 64
            buf += struct.pack("<QQqqiiiiii", src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), R, C, dst.shape[1], tile0,
                               tiles_c, 0)
            tile0 += tiles_c * ((dst.shape[1] + 63) // 64)
        dev = next(iter(self._tr.values()))[0].device
        tab = torch.frombuffer(buf, dtype=torch.uint8).to(dev)
        if self._tr_table_old is not None:
            _KEEPALIVE.append(self._tr_table_old)       # an earlier capture's transpose node still reads its own table
        self._tr_table = (tab, len(self._tr), tile0)
        self._tr_table_old = tab
        return True

    def _refresh_transposed(self):
        """One launch rebuilds every transposed shadow from the (current) bf16 mirror; False = not possible right now."""
        if not self.prepare():
            return False
        tab, n, tiles = self._tr_table
        _lib.call("vlni_transpose_batched", _DT[self.arena[1].dtype], tab.data_ptr(), n, tiles, _st())
        for k in self._tr:
            e = self._c[k]
            self._c[k] = (e[0], e[1], e[2], self.opt_epoch, e[4])
        return True

    def invalidate(self, optimizer_step=False):
        """optimizer_step=True: the fused AdamW kernel just rewrote the float32 arena AND its bf16 mirror (raw pointers: tensor
        version counters do not move). Otherwise: parameters changed behind autograd's back, rebuild everything lazily."""
        if optimizer_step and self.arena is not None:
            self.opt_epoch += 1
        else:
            self.epoch += 1

    def _mirror(self, params):
        """bf16 view of `params` inside the arena mirror (they must be packed neighbours inside the arena), else None."""
        if self.arena is None:
            return None
        flat_p, flat_b = self.arena
        v = _packed_param(params) if len(params) > 1 else params[0].detach()
        if v is None or not v.is_contiguous():
            return None
        off = (v.data_ptr() - flat_p.data_ptr()) // 4
        if v.dtype != torch.float32 or off < 0 or off + v.numel() > flat_p.numel() or (v.data_ptr() - flat_p.data_ptr()) % 4:
            return None
        return v, flat_b[off:off + v.numel()].view(v.shape)

    def get(self, params, dtype, transposed=False):
        key = (id(params[0]), len(params), dtype, transposed)
        hit = self._c.get(key)
        ver = (self.epoch, params[0]._version, params[0].data_ptr(), sum(p._version for p in params[1:]))
        if hit is not None and hit[4]() is not params[0]:
            hit = None                 # a dead parameter's id was recycled
        if hit is not None and hit[0] == ver and (hit[2] or hit[3] == self.opt_epoch):
            return hit[1]
        if hit is not None and hit[0] == ver and key in self._tr and self._refresh_transposed():
            return hit[1]              # steady state: only the optimizer moved the parameters, the mirror is current
        if hit is not None and hit[0] != ver and len(hit) > 5 and BATCH_SHADOWS and self._refresh_plain():
            hit = self._c.get(key)     # another optimizer (torch.optim) stepped: every stale 16-bit copy re-cast in one launch
            if hit is not None and hit[0] == ver:
                return hit[1]
        live = False                   # True: `t` aliases memory the optimizer keeps current (never stale within an epoch)
        with torch.no_grad():
            for p in params:
                _chk(p, "parameter")
            mir = self._mirror(params) if (self.arena is not None and dtype == self.arena[1].dtype) else None
            if mir is not None and (hit is None or hit[0] != ver):
                # (re)fill this slice of the mirror once per epoch; afterwards the optimizer kernel maintains it
                _lib.call("vlni_cast", _DT[torch.float32], _DT[dtype], mir[0].data_ptr(), mir[1].data_ptr(), mir[0].numel(), _st())
            if dtype == torch.float32 and not transposed:
                t = params[0].detach() if len(params) == 1 else _packed_param(params)
                live = t is not None
                if t is None:
                    t = torch.cat([p.detach() for p in params], 0)
            elif mir is not None and not transposed:
                t, live = mir[1], True
            else:
                src = mir[1] if mir is not None else \
                    (params[0].detach() if len(params) == 1 else torch.cat([p.detach() for p in params], 0))
                if src.dim() == 1:
                    t = cast(src, dtype)
                elif transposed:
                    t = transpose_pad(src, src.shape[0], dtype)
                    if mir is not None:
                        self._tr[key], self._tr_table = (mir[1], t), None
                else:
                    t = cast(src, dtype)
        if not any(p.requires_grad for p in params):
            live = True                # frozen parameters: no optimizer step ever touches them
        if mir is None and (dtype in H16 or dtype == torch.float32) and not live \
                and all(p.dtype == torch.float32 and p.dim() == params[0].dim() <= 2 for p in params):
            # a plain float32 parameter's own 16-bit copy (or the float32 row-pack of several: Q | K | V biases): _refresh_plain() keeps it current in place
            self._c[key] = (ver, t, live, self.opt_epoch, weakref.ref(params[0]), tuple(weakref.ref(p) for p in params))
        else:
            self._c[key] = (ver, t, live, self.opt_epoch, weakref.ref(params[0]))
        return t

    def _refresh_plain(self):
        """Re-casts, in ONE launch per dtype, every cached 16-bit copy of plain (not arena-resident) float32 parameters whose parameters
        changed since it was made - the ~200 cast / transpose launches per iteration that an unchanged agent's own optimizer.step() used to
        cost (each copy was rebuilt on its first use, with a new allocation). Copies are overwritten in place."""
        if torch.cuda.is_current_stream_capturing() or _ghost():
            return False
        import struct
        per = {}
        for key, e in self._c.items():
            if len(e) <= 5:
                continue
            ps = [r() for r in e[5]]
            if any(p is None for p in ps) or e[4]() is not ps[0]:
                continue
            ver = (self.epoch, ps[0]._version, ps[0].data_ptr(), sum(p._version for p in ps[1:]))
            if ver == e[0]:
                continue
            t, transposed = e[1], key[3]
            R = sum(p.shape[0] for p in ps) if ps[0].dim() == 2 else 1
            C = ps[0].shape[-1]
            want = (C, R) if transposed else ((R, C) if ps[0].dim() == 2 else (sum(p.shape[0] for p in ps),))
            if tuple(t.shape) != want or not t.is_contiguous() or any(not p.is_contiguous() for p in ps):
                continue
            per.setdefault(t.dtype, []).append((key, e, ps, ver, transposed))
        if not per:
            return False
        for dt, lst in per.items():
            sig = tuple((p.data_ptr(), e[1].data_ptr()) for _, e, ps, _, _ in lst for p in ps)
            tab = self._plain_tables.get((dt, sig))
            if tab is None:
                buf, tile0 = bytearray(), 0
                for _, e, ps, _, transposed in lst:
                    t, off = e[1], 0
                    for p in ps:
                        es = t.element_size()
                        if p.dim() == 1:
                            r, c, lds, ldd, dst = 1, p.shape[0], p.shape[0], p.shape[0], t.data_ptr() + es * off
                            off += p.shape[0]
                        elif transposed:
                            r, c, lds, ldd, dst = p.shape[0], p.shape[1], p.stride(0), t.stride(0), t.data_ptr() + es * off
                            off += p.shape[0]
                        else:
                            r, c, lds, ldd, dst = p.shape[0], p.shape[1], p.stride(0), t.stride(0), t.data_ptr() + es * off * t.stride(0)
                            off += p.shape[0]
                        tiles_c = (c + 63) // 64
                        buf += struct.pack("<QQqqiiiiii", p.data_ptr(), dst, lds, ldd, r, c, 1 if (transposed and p.dim() == 2) else 0, tile0, tiles_c, 0)
                        tile0 += tiles_c * ((r + 63) // 64)
                if len(self._plain_tables) > 64:
                    self._plain_tables.clear()
                tab = self._plain_tables[(dt, sig)] = (torch.frombuffer(buf, dtype=torch.uint8).to(lst[0][1][1].device), len(buf) // 56, tile0)
            _lib.call("vlni_shadow_refresh", _DT[dt], tab[0].data_ptr(), tab[1], tab[2], _st())
            for key, e, _, ver, _ in lst:
                self._c[key] = (ver,) + tuple(e[1:])
        return True


    def current_for_replay(self, ids=None):
        """For captured graphs that hold the copies' ADDRESSES (vln_imagine_amd/graphed.py): re-casts every stale refreshable copy in place and
        says whether every cached copy (of the parameters whose id() is in `ids`; None = all) is now current - False when one could only be
        rebuilt by a new allocation (the graphs are then dropped)."""
        self._refresh_plain()
        for key, e in self._c.items():
            if e[2]:
                continue                                   # aliases memory that is current by construction
            p0 = e[4]()
            if p0 is None or (ids is not None and id(p0) not in ids):
                continue
            if len(e) > 5:
                ps = [r() for r in e[5]]
                if any(p is None for p in ps):
                    continue
                ver = (self.epoch, ps[0]._version, ps[0].data_ptr(), sum(p._version for p in ps[1:]))
                if ver != e[0]:
                    return False
            elif (self.epoch, p0._version, p0.data_ptr()) != e[0][:3]:
                return False
        return True


BATCH_SHADOWS = os.environ.get("VLNI_BATCH_SHADOWS", "1") == "1"
SHADOWS = ShadowCache()


def _w(params, dtype, transposed=False):
    """Operand form of a (possibly row-packed) parameter. transposed=True is the dgrad operand: W^T [in, out] for the NT kernels,
    or - bf16 with NN_DGRAD - a WT handle that the GEMM front-ends turn into the untransposed weight (KN, transposing-read kernels,
    no copy to rebuild) or the W^T copy depending on the launch's row count."""
    if transposed and NN_DGRAD and dtype in H16 and params[0].dim() == 2:
        out_f = sum(p.shape[0] for p in params)
        if out_f % 64 == 0 and out_f >= 192 and params[0].shape[1] % 8 == 0:
            return WT(params, dtype)
    return SHADOWS.get(params, dtype, transposed)


def _split_rows(t, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(t[o:o + s])
        o += s
    return out


# Parameters that train.FlatTrainer marked (`_vlni_direct`) get their gradients accumulated by the kernels straight into the
# pre-allocated `.grad` arena views (wgrad atomics, colsum, LayerNorm dgamma/dbeta, embedding scatter) and the
# autograd functions return None for them: no zero-filled temporaries, no AccumulateGrad add kernels. The mark is per
# parameter, so a second model in the same process keeps plain autograd accumulation.
def _direct(*params, queue=False):
    if params and getattr(params[0], "_vlni_auto", None) is not None:
        _session(params)
    ok = all(p is not None and p.grad is not None and getattr(p, "_vlni_direct", False) for p in params)
    if ok and not queue and GRADS.arena is not None:
        # a kernel is about to ADD into these gradients: one that the reduction also writes (GradArena) is zeroed now if the arena's
        # zero fill skipped it, and kept out of the stored ranges from here on
        for p in params:
            if getattr(p, "_vlni_queued", False):
                GRADS.mixed(p.grad)
    return ok


def _packed_grad(params):
    """One [sum rows, ...] view over the .grad of parameters that sit back to back in the arena, else None."""
    g0 = params[0].grad
    ptr = g0.data_ptr()
    st0 = g0.untyped_storage().data_ptr()
    for p in params:
        # neighbours in ONE allocation (separately allocated gradients can happen to sit back to back)
        if p.grad.data_ptr() != ptr or not p.grad.is_contiguous() or p.grad.untyped_storage().data_ptr() != st0:
            return None
        ptr += p.grad.numel() * 4
    rows = sum(p.shape[0] for p in params)
    return torch.as_strided(g0, (rows,) + tuple(g0.shape[1:]), g0.stride(), g0.storage_offset())


def _wgrad_to(params, dy, x):
    """Weight gradient of a (possibly row-packed) projection. Returns per-parameter grads or Nones (direct mode)."""
    if _direct(*params):
        _added(params, ())
        view = _packed_grad(params) if len(params) > 1 else params[0].grad
        if view is not None:
            wgrad(dy, x, out=view)
        else:
            tmp = wgrad(dy, x)
            for p, t in zip(params, _split_rows(tmp, [q.shape[0] for q in params])):
                p.grad.add_(t)
        return (None,) * len(params)
    if len(params) == 1:
        return (wgrad(dy, x),)
    return tuple(_split_rows(wgrad(dy, x), [q.shape[0] for q in params]))


def _bgrad_to(params, dy):
    if _direct(*params):
        _added(params, ())
        view = _packed_grad(params) if len(params) > 1 else params[0].grad
        if view is not None:
            colsum(dy, out=view)
        else:
            tmp = colsum(dy)
            for p, t in zip(params, _split_rows(tmp, [q.shape[0] for q in params])):
                p.grad.add_(t)
        return (None,) * len(params)
    if len(params) == 1:
        return (colsum(dy),)
    return tuple(_split_rows(colsum(dy), [q.shape[0] for q in params]))


# Deferred weight gradients (parameters marked `_vlni_defer` by train.FlatTrainer): the (dY, X) pairs of a parameter are
# queued during backward and reduced by ONE grouped transposing-read GEMM per parameter at flush time, i.e. one launch
# with a T-times longer reduction instead of T short split-K launches. 288 GB of HBM make keeping dY alive free.
RESERVE_CUS = 0          # > 0 (train.FlatTrainer with a gradient exchange): one-round weight-gradient launches leave this many CUs free
TN_BIG = True            # 256 x 256 tiles for episode-long reductions (+10..26 % there, tools/tn_probe.py)
TN_VARIANT = 5          # LDS-DMA 2-stage, 8 waves: fastest of the five on every episode-level shape (tools/tn_probe.py)
_WQ = {}

STORE_PARTS = os.environ.get("VLNI_STORE_PARTS", "1") == "1"
PART_STORE, PART_NOWRITE, PART_SUMSQ = 1 << 16, 1 << 17, 1 << 18        # include/vlni.h VLNI_PART_*
SUMSQ_SLOTS = 64


class GradArena:
    """Which parts of train.FlatTrainer's gradient arena the batched partial reduction of flush_wgrads() will WRITE this step (store mode).

    The reduction used to add the row-split slabs onto a zero-filled arena: a 4 B / parameter fill, a 4 B / parameter read of the zeros
    and a separate 4 B / parameter pass for the gradient norm, every step. The ranges one step reduced are what the next step's
    begin() leaves out of the zero fill (`pending`); the flush stores into exactly those (no read), adds onto anything else, zero-fills
    what is left of `pending` (a parameter without a gradient this step) and - one rank, first reduction of the step - also leaves each
    stored range's sum of squares in the trainer's accumulator, so step() only adds the ranges no reduction wrote.
    Anything else that adds into a pending range (an unsplit weight-gradient launch's float atomics, an immediate weight gradient)
    goes through touch(), which zero-fills it first. begin() and the flush of one step are either both eager or both in the same
    captured graph, so a replayed step is consistent with itself whatever ran in between."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.arena = None          # the flat float32 gradient tensor
        self.known = {}            # element offset -> elements: what the last completed step's reductions stored or added
        self.pending = {}          # left out of this step's zero fill, not written yet
        self.seen = {}             # this step's reduction destinations (the next step's `known`)
        self.sumsq = None          # accumulator the stored ranges' sums of squares go to (None: no folding this step)
        self.slots = None
        self.folded = {}           # ranges whose sum of squares is in it
        self.fold_ok = False
        self._starts = None
        self._tables = {}
        self._prebuilt = None
        self.never = {}            # offset -> elements: gradients that something besides the reduction also adds into (never stored)

    def _table(self, kind, ranges, flags):
        """Device table of reduce_parts entries without partials over element ranges (zero fill / sum of squares)."""
        key = (kind, tuple(ranges))
        tab = self._tables.get(key)
        if tab is None:
            arr = np.zeros((len(ranges),), _PART_DT)
            blk, base = 0, self.arena.data_ptr()
            for i, (o, n) in enumerate(ranges):
                arr[i] = (base + 4 * o, 0, n // 4, 0, flags, blk)
                blk += -(-(n // 4) // 1024)
            tab = self._tables[key] = (_dev_table(arr, self.arena.device), len(ranges), blk)
        return tab

    def _gaps(self, covered):
        """Complement of `covered` ({offset: elements}, disjoint) in the arena, as (offset, elements) runs."""
        out, at = [], 0
        for o in sorted(covered):
            if o > at:
                out.append((at, o - at))
            at = max(at, o + covered[o])
        if at < self.arena.numel():
            out.append((at, self.arena.numel() - at))
        return out

    def begin(self, arena, sumsq=None):
        """Start of a step (FlatTrainer.zero_grad): zero the arena except what the last step's reductions covered. sumsq: float32
        [32 (1 + SUMSQ_SLOTS)] - element 0 is the sum the optimizer reads, the slots follow 32 floats apart (vlni_reduce_parts_sq)."""
        if self.arena is None or self.arena.data_ptr() != arena.data_ptr() or self.arena.numel() != arena.numel():
            if self.arena is not None:
                self.end()               # another trainer's arena: nothing of it stays un-zeroed
            self.reset()
            self.arena = arena
        elif self.seen:
            self.known = self.seen
        self.seen, self.folded, self._starts = {}, {}, None
        self.sumsq, self.fold_ok = sumsq, sumsq is not None
        if sumsq is not None:
            assert sumsq.numel() == 32 * (1 + SUMSQ_SLOTS)
            self.slots = sumsq[32:]
            sumsq.zero_()
        if not (STORE_PARTS and WGRAD_PARTS and self.known):
            self.pending = {}
            arena.zero_()
            return
        self.pending = dict(self.known)
        gaps = self._gaps(self.pending)
        if gaps:
            tab = self._table("zero", gaps, PART_STORE)
            _lib.call("vlni_reduce_parts_sq", tab[0].data_ptr(), tab[1], tab[2], None, 1, _st())

    def end(self):
        """After the optimizer step: nothing is pending or folded until the next begin()."""
        if self.pending:                 # a step without a flush (cannot happen through FlatTrainer.step): keep the arena consistent
            for o, n in list(self.pending.items()):
                self.arena[o:o + n].zero_()
            self.pending = {}
        self.sumsq, self.slots, self.fold_ok, self.folded = None, None, False, {}

    def _overlapping(self, o, n):
        if self._starts is None:             # sorted once per step: `pending` only shrinks until the next begin()
            self._starts = sorted(self.pending)
        st = self._starts
        i = bisect.bisect_right(st, o) - 1
        hit = []
        if i >= 0 and st[i] in self.pending and st[i] + self.pending[st[i]] > o:
            hit.append(st[i])
        i += 1
        while i < len(st) and st[i] < o + n:
            if st[i] in self.pending:
                hit.append(st[i])
            i += 1
        return hit

    def offset(self, t):
        return (t.data_ptr() - self.arena.data_ptr()) // 4

    def touch(self, t):
        """`t` (a view of the arena) is about to be accumulated into by something other than the reduction."""
        if not self.pending or self.arena is None:
            return
        o = self.offset(t)
        if not 0 <= o < self.arena.numel():
            return
        n = t.numel() if t.is_contiguous() else t.stride(0) * t.shape[0]
        for h in self._overlapping(o, n):
            self.arena[h:h + self.pending.pop(h)].zero_()
        self.fold_ok = self.fold_ok and not any(f < o + n and f + m > o for f, m in self.folded.items())

    def mixed(self, t):
        """`t`'s gradient is both queued for the reduction and added into directly (the same projection called once with operands the
        grouped launch takes and once without): it stays in the zero-filled, added-onto part of the arena for good."""
        if self.arena is None:
            return
        o = self.offset(t)
        if 0 <= o < self.arena.numel() and o not in self.never:
            self.never[o] = t.numel() if t.is_contiguous() else t.stride(0) * t.shape[0]
            self.touch(t)

    def will_store(self, dst_ptr, n):
        """Whether the next step stores into this destination if it has this step's structure (flush_wgrads pre-builds that step's table)."""
        return self.arena is not None and self.seen.get((dst_ptr - self.arena.data_ptr()) // 4) == n

    def claim(self, dst_ptr, n):
        """Mode flags of a reduction entry for [dst_ptr, + n floats): stored into (left out of the zero fill, first write) or added onto."""
        if self.arena is None:
            return 0
        o = (dst_ptr - self.arena.data_ptr()) // 4
        if not 0 <= o < self.arena.numel():
            return 0
        if self.never and any(f < o + n and f + m > o for f, m in self.never.items()):
            for h in self._overlapping(o, n):
                self.arena[h:h + self.pending.pop(h)].zero_()
            return 0
        self.seen[o] = n
        if self.pending.get(o) == n:
            del self.pending[o]
            if self.fold_ok:
                self.folded[o] = n
                return PART_STORE | PART_SUMSQ
            return PART_STORE
        for h in self._overlapping(o, n):                       # a differently shaped view of a skipped range (rare): zero it, then add
            self.arena[h:h + self.pending.pop(h)].zero_()
        if any(f < o + n and f + m > o for f, m in self.folded.items()):
            self.fold_ok = False                                # second reduction onto a folded range: the folded sum is stale
        return 0

    def leftovers(self, lo=None, hi=None):
        """Pending ranges (inside [lo, hi) byte addresses) that got no gradient this step: (offset, elements) to zero-fill."""
        base = self.arena.data_ptr() if self.arena is not None else 0
        out = []
        for o in sorted(self.pending):
            if lo is None or lo <= base + 4 * o < hi:
                out.append((o, self.pending.pop(o)))
        return out

    def prebuild(self):
        """After an eager flush: the device tables the NEXT step needs if it has this step's structure (its zero fill around `seen`, the
        sum of squares outside it), so that a capture of that step finds them made - a table upload cannot be recorded into a graph."""
        if self.arena is None or not (STORE_PARTS and WGRAD_PARTS) or not self.seen or self._prebuilt == self.seen:
            return
        gaps = self._gaps(self.seen)
        if gaps:
            self._table("zero", gaps, PART_STORE)
            self._table("sumsq", gaps, PART_NOWRITE | PART_SUMSQ)
        self._prebuilt = dict(self.seen)

    def finish_sumsq(self):
        """step(): True when the accumulator holds the stored ranges' sums of squares; adds the rest of the arena to it."""
        if not (self.fold_ok and self.folded and self.sumsq is not None) or self.pending:
            return False
        gaps = self._gaps(self.folded)
        if gaps:
            tab = self._table("sumsq", gaps, PART_NOWRITE | PART_SUMSQ)
            _lib.call("vlni_reduce_parts_sq", tab[0].data_ptr(), tab[1], tab[2], self.slots.data_ptr(), SUMSQ_SLOTS, _st())
        _lib.call("vlni_sumsq_fold", self.slots.data_ptr(), SUMSQ_SLOTS, self.sumsq.data_ptr(), _st())
        return True


GRADS = GradArena()


_TN_BEST = {}
# 256 x 256 weight-gradient tiles: 6 = two 64-KiB stages, 7 = ring of 32-row half-steps, 8 = the ring with its wave rows one barrier apart (round 6)
_RING_VARIANTS = (6, 7, 8) if os.environ.get("VLNI_TN_RING8", "1") == "1" else (6, 7)


SMALL_TABLE_SCATTER = os.environ.get("VLNI_SMALL_SCATTER", "1") == "1"
WGRAD_PARTS = os.environ.get("VLNI_WGRAD_PARTS", "1") == "1"   # row splits write partial gradients with plain stores + ONE batched reduction per
                                                              # step instead of float atomics (~1.3 TB/s on this chip, 30-50 % of a launch)
_PART_BUFS = {}       # (gradient address, N, K) -> workspace: [splits][N][K] slabs, then [splits][N] column sums
_KEEPALIVE = []       # workspaces / device tables that were replaced by larger ones: captured hipGraphs hold RAW pointers to them, and
                      # torch.cuda.graph() empties the allocator cache on entry - a freed one would be unmapped under an older graph
                      # (memory access fault on its next replay). Dropped by train.FlatTrainer.close().
_PART_TABLES = {}     # signature of a flush -> (device table, entries, blocks)
_PART_DT = np.dtype([("dst", "<u8"), ("part", "<u8"), ("n4", "<i8"), ("stride4", "<i8"), ("split", "<i4"), ("blk0", "<i4")])


def _eff_split(nmt, split):
    """Row splits a grouped launch really produces (include/vlni.h)."""
    per = -(-nmt // split)
    return -(-nmt // per), per


def _parts_ok(variant, nmt, split):
    eff, per = _eff_split(nmt, split)
    return WGRAD_PARTS and eff > 1 and variant >= 2 and per >= 3


def _tn_choice(n, pa, pb, pm, N, K, nmt, dev, dtid=BF16):
    """(kernel variant, row split) of a grouped weight-gradient launch. Which of the 128x128 LDS-DMA kernel (more, smaller blocks)
    and the 256x256 tiles (half the operand traffic, one block per CU) wins, and at which split, depends on the output size and on
    the length of the reduction (tools/tn_probe.py): the first launch of a (N, K, rows) class times the candidates on a scratch
    output and the winner is cached. Split candidates are timed in the mode they will run in (partials + their share of the
    batched reduction, priced at 4 TB/s; atomics otherwise)."""
    default = (TN_VARIANT, max(1, min(8, nmt // 8)))
    if not AUTOTUNE:
        return default
    key = (N, K, nmt // 16)
    best = _TN_BEST.get(key)
    if best is not None or torch.cuda.is_current_stream_capturing():
        return best or default
    t128 = ((N + 127) // 128) * ((K + 127) // 128)
    cands = {(TN_VARIANT, max(1, min(s, nmt // 3))) for s in (2, 3, 4, 6, 8, 12, 16, max(1, round((512 - 2 * RESERVE_CUS) / t128)))}
    if TN_BIG and N >= 256 and K >= 256 and nmt >= 32:
        t256 = ((N + 255) // 256) * ((K + 255) // 256)
        s6 = max(1, min(nmt // 4, round((252 - RESERVE_CUS) / t256)))
        cands |= {(v, sp) for v in _RING_VARIANTS for sp in (s6, max(1, s6 // 2), max(1, (3 * s6) // 4), min(nmt // 4, 2 * s6))}
    smax = max(_eff_split(nmt, sp)[0] for _, sp in cands)
    scratch = torch.zeros((smax * (N * K + N),), dtype=torch.float32, device=dev)
    timed = []
    for v, sp in sorted(cands):
        if _parts_ok(v, nmt, sp):
            eff = _eff_split(nmt, sp)[0]
            args = ("vlni_gemm_tn_h16_grouped_part", dtid, n, pa, pb, pm, N, K, scratch.data_ptr(), N * K, N, K,
                    scratch.data_ptr() + 4 * smax * N * K, sp, v, _st())
            extra = (eff + 2) * (N * K + N) * 4 / 4e9            # ms: its share of the batched reduction
        else:
            args = ("vlni_gemm_tn_h16_grouped_v", dtid, n, pa, pb, pm, N, K, scratch.data_ptr(), K, N, K, scratch.data_ptr() + 4 * N * K,
                    sp, v, _st())
            extra = 0.0
        _lib.call(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(*args); _lib.call(*args)
        e1.record()
        e1.synchronize()
        timed.append((e0.elapsed_time(e1) / 2 + extra, v, sp))
    _, v, sp = min(timed)
    _TN_BEST[key] = (v, sp)
    return v, sp


BATCH_WGRADS = int(os.environ.get("VLNI_BATCH_WGRADS", "1"))      # 0 off, 1 on, 2 = also the wide multi-segment gradients (probe)
_TNB_BEST = {}


def _tn_batch_choice(P, n, pa, pb, pm, N, K, nmt_p, dev, dtid):
    """(kernel variant, row splits per gradient) of a BATCHED weight-gradient launch: P gradients of one shape with the same (dY, X)
    segment lengths each (n segments in all, nmt_p 64-row tiles per gradient). The splits per gradient must divide nmt_p so that no
    block's rows cross into the next gradient."""
    divs = [d for d in range(1, nmt_p + 1) if nmt_p % d == 0 and nmt_p // d >= 4]
    default = (TN_VARIANT, divs[min(len(divs) - 1, 1)] if divs else 1)
    if not AUTOTUNE:
        return default
    key = (N, K, nmt_p, P, n, dtid)
    best = _TNB_BEST.get(key)
    if best is not None or torch.cuda.is_current_stream_capturing():
        return best or default
    t128, t256 = -(-N // 128) * -(-K // 128), -(-N // 256) * -(-K // 256)
    cands = set()
    for v, tiles in ((TN_VARIANT, t128),) + (tuple((v_, t256) for v_ in _RING_VARIANTS) if TN_BIG and N >= 256 and K >= 256 else ()):
        want = max(1.0, ((512 if v == TN_VARIANT else 250) - (2 if v == TN_VARIANT else 1) * RESERVE_CUS) / (P * tiles))   # ~ one round of blocks
        near = sorted(divs, key=lambda d: abs(d - want))[:3]
        cands |= {(v, d) for d in near}
    smax = max(sp for _, sp in cands) * P
    scratch = torch.zeros((smax * (N * K + N),), dtype=torch.float32, device=dev)
    timed = []
    for v, sp in sorted(cands):
        if not _parts_ok(v, P * nmt_p, P * sp):
            continue
        args = ("vlni_gemm_tn_h16_grouped_part", dtid, n, pa, pb, pm, N, K, scratch.data_ptr(), N * K, N, K,
                scratch.data_ptr() + 4 * smax * N * K, P * sp, v, _st())
        _lib.call(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(*args); _lib.call(*args)
        e1.record()
        e1.synchronize()
        timed.append((e0.elapsed_time(e1) / 2 + P * (sp + 2) * (N * K + N) * 4 / 4e9, v, sp))
    if not timed:
        return None
    _, v, sp = min(timed)
    _TNB_BEST[key] = (v, sp)
    return v, sp


TN_MAXSEG = 32          # csrc/gemm_impl.inc: row segments one grouped launch takes


def _flush_batch(members, entries):
    """ONE grouped launch for several gradients of the same shape whose (dY, X) segments have the same lengths (the text encoder's
    layers: 9 x {QKV, O, FFN-in, FFN-out} with one segment each; the cross-modal layers' 768 x 768 projections with one segment per
    step): the launch's row-tile space is the gradients' segments back to back, its P x s row splits fall on gradient boundaries,
    split z writes slab z, and the batched reduction adds slabs [p s, (p + 1) s) into gradient p. Per gradient these launches were
    9-144 tiles x 3-21 splits (350-500 TF/s over the classes)."""
    P = len(members)
    d0, x0 = members[0][3][0]
    N, K, dtid, dev = d0.shape[1], x0.shape[1], _dt(d0), d0.device
    pairs = [pr for m in members for pr in m[3]]             # member after member: every gradient's segments stay together
    n = len(pairs)
    pa = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    pb = (ctypes.c_void_p * n)(*[x.data_ptr() for _, x in pairs])
    pm = (ctypes.c_int * n)(*[d.shape[0] for d, _ in pairs])
    nmt_p = sum((d.shape[0] + 63) // 64 for d, _ in members[0][3])
    choice = _tn_batch_choice(P, n, pa, pb, pm, N, K, nmt_p, dev, dtid)
    if choice is None or not _parts_ok(choice[0], P * nmt_p, P * choice[1]):
        return False
    variant, sp = choice
    tot = P * sp
    key = (members[0][1].data_ptr(), N, K, "batch", P)
    buf = _PART_BUFS.get(key)
    if buf is None or buf.numel() < tot * (N * K + N):
        if buf is not None:
            _KEEPALIVE.append(buf)
        buf = _PART_BUFS[key] = torch.empty((tot * (N * K + N),), dtype=torch.float32, device=dev)
    cpart = buf.data_ptr() + 4 * tot * N * K
    _lib.call("vlni_gemm_tn_h16_grouped_part", dtid, n, pa, pb, pm, N, K, buf.data_ptr(), N * K, N, K, cpart, tot, variant, _st())
    for i, (_, wv, bv, _) in enumerate(members):
        entries.append((wv.data_ptr(), buf.data_ptr() + 4 * i * sp * N * K, N * K // 4, N * K // 4, sp))
        entries.append((bv.data_ptr(), cpart + 4 * i * sp * N, N // 4, N // 4, sp))
    return True


_PINNED = []          # [pinned uint8 tensor, bytes used]: staging for tables that have to be uploaded while a stream is capturing


def reserve_staging(nbytes=1 << 20):
    """Pinned staging for _dev_table's in-capture uploads; FlatTrainer calls this at construction (pinning is not allowed mid-capture)."""
    if not _PINNED or _PINNED[-1][0].numel() - _PINNED[-1][1] < nbytes // 2:
        _PINNED.append([torch.empty(nbytes, dtype=torch.uint8).pin_memory(), 0])


def _dev_table(arr, dev):
    """A host table (numpy) on the device. Outside a capture: an ordinary copy. While the stream is capturing (a launch table that no
    eager step has needed yet): through a slice of the pinned staging buffer that is never written again, as a memcpy node of the graph."""
    host = torch.from_numpy(arr.view(np.uint8).reshape(-1))
    if not torch.cuda.is_current_stream_capturing():
        return host.to(dev)
    n = host.numel()
    if not _PINNED or _PINNED[-1][0].numel() - _PINNED[-1][1] < n:
        raise RuntimeError(f"a {n}-byte launch table is needed inside a hipGraph capture and the pinned staging buffer is full or missing "
                           "(ops.reserve_staging() before capturing; one more eager warm-up step also avoids the upload)")
    buf, used = _PINNED[-1]
    _PINNED[-1][1] = used + (n + 63) // 64 * 64
    buf[used:used + n].copy_(host)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.call("vlni_upload", out.data_ptr(), buf.data_ptr() + used, n, _st())
    return out


def _part_table(entries, dev):
    arr = np.zeros((len(entries),), _PART_DT)
    blk = 0
    for i, (dst, part, n4, stride4, eff) in enumerate(entries):
        arr[i] = (dst, part, n4, stride4, eff, blk)
        blk += -(-n4 // 1024)
    return _dev_table(arr, dev), len(entries), blk


def flush_wgrads(lo=None, hi=None, queue=None, store=None):
    """Runs the queued weight/bias gradient reductions (must precede any read of the .grad arena). With (lo, hi): only the
    gradients whose address lies in [lo, hi), in address order (train.FlatTrainer's flush -> all-reduce pipeline).
    queue / store: another queue than the trainer's (the end-of-backward flush of an agent model's GradSession) and the set of its
    destination addresses that hold no gradient yet (written, not added to)."""
    entries, dev = [], None
    wq = _WQ if queue is None else queue
    if lo is None:
        todo = list(wq)
    else:
        todo = sorted(k for k in wq if lo <= k < hi)
    items = [(key,) + tuple(wq.pop(key)) for key in todo]
    if BATCH_WGRADS and WGRAD_PARTS and len(items) > 1:
        groups, rest = {}, []
        for it in items:
            _, wv, bv, segs = it
            d, x = segs[0]
            # several segments per gradient (the steps of an episode): only the small outputs, whose own launches need 14-21 row
            # splits to fill the chip; the 2304- and 3072-wide ones already run at 1.2-1.3 PFLOP/s on their own
            if 2 * len(segs) <= TN_MAXSEG and wv.is_contiguous() and bv.is_contiguous() and d.shape[1] % 4 == 0 \
                    and sum(dd.shape[0] for dd, _ in segs) >= 256 and (len(segs) == 1 or d.shape[1] * x.shape[1] <= 768 * 768 or BATCH_WGRADS == 2):
                groups.setdefault((d.shape[1], x.shape[1], tuple(dd.shape[0] for dd, _ in segs), d.dtype), []).append(it)
            else:
                rest.append(it)
        for (_, _, lens, _), members in groups.items():
            per = max(2, TN_MAXSEG // len(lens))                     # gradients per launch: their segments must fit one launch
            for c in range(0, len(members), per):
                chunk = members[c:c + per]
                if len(chunk) >= 2 and _flush_batch(chunk, entries):
                    dev = chunk[0][1].device
                else:
                    rest.extend(chunk)
        items = rest
    for key, wv, bv, segs in items:
        dev = wv.device
        N, K = segs[0][0].shape[1], segs[0][1].shape[1]
        plans, tot = [], 0                 # a gradient with more than 16 segments (an episode of > 8 steps through a shared module)
        for c in range(0, len(segs), 16):  # takes several grouped launches; their row splits all land in ONE slab workspace
            chunk = segs[c:c + 16]
            n = len(chunk)
            pa = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in chunk])
            pb = (ctypes.c_void_p * n)(*[x.data_ptr() for _, x in chunk])
            pm = (ctypes.c_int * n)(*[d.shape[0] for d, _ in chunk])
            nmt = sum((d.shape[0] + 63) // 64 for d, _ in chunk)
            dtid = _dt(chunk[0][0])
            variant, split = _tn_choice(n, pa, pb, pm, N, K, nmt, wv.device, dtid)
            parts = _parts_ok(variant, nmt, split) and wv.is_contiguous() and bv.is_contiguous() and N % 4 == 0
            eff = _eff_split(nmt, split)[0] if parts else 0
            plans.append((n, pa, pb, pm, variant, split, tot if parts else -1, dtid))
            tot += eff
        buf = None
        if tot:
            key = (wv.data_ptr(), N, K)                      # one workspace per gradient, grown to the largest split count seen
            buf = _PART_BUFS.get(key)
            if buf is None or buf.numel() < tot * (N * K + N):
                if buf is not None:
                    _KEEPALIVE.append(buf)     # a captured step (another shape bucket's graph) may still write its partials there
                buf = _PART_BUFS[key] = torch.empty((tot * (N * K + N),), dtype=torch.float32, device=wv.device)
            cpart = buf.data_ptr() + 4 * tot * N * K         # [tot][N] column-sum partials behind the [tot][N][K] slabs
        if store is not None and any(pl[6] < 0 for pl in plans):          # float atomics add: a destination without a gradient starts at zero
            for t in (wv, bv):
                if t.data_ptr() in store:
                    t.zero_()
                    store.discard(t.data_ptr())
        for n, pa, pb, pm, variant, split, z0, dtid in plans:
            if z0 >= 0:
                _lib.call("vlni_gemm_tn_h16_grouped_part", dtid, n, pa, pb, pm, N, K, buf.data_ptr() + 4 * z0 * N * K, N * K, N, K,
                          cpart + 4 * z0 * N, split, variant, _st())
            else:                                            # unsplit (or register-staged) launch: float atomics straight into the arena
                if store is None:
                    GRADS.touch(wv); GRADS.touch(bv)
                _lib.call("vlni_gemm_tn_h16_grouped_v", dtid, n, pa, pb, pm, N, K, wv.data_ptr(), wv.stride(0), N, K, bv.data_ptr(),
                          split, variant, _st())
        if tot:
            # exactly ONE reduction entry per destination: reduce_parts_kernel's read-modify-write of dst is not atomic
            entries.append((wv.data_ptr(), buf.data_ptr(), N * K // 4, N * K // 4, tot))
            entries.append((bv.data_ptr(), cpart, N // 4, N // 4, tot))
    if store is not None:
        entries = [(dst, part, n4, stride4, eff | (PART_STORE if dst in store else 0)) for dst, part, n4, stride4, eff in entries]
    elif GRADS.arena is not None:
        # store mode (GradArena): a destination the step's zero fill left out is written, not added to; what is left of those gets zeros
        entries = [(dst, part, n4, stride4, eff | GRADS.claim(dst, 4 * n4)) for dst, part, n4, stride4, eff in entries]
        base = GRADS.arena.data_ptr()
        entries += [(base + 4 * o, 0, n // 4, 0, PART_STORE) for o, n in GRADS.leftovers(lo, hi)]
        dev = GRADS.arena.device
    if entries:
        sig = tuple(entries)
        tab = _PART_TABLES.get(sig)
        if tab is None:
            tab = _PART_TABLES[sig] = _part_table(sig, dev)
        sq = GRADS.slots if store is None and GRADS.fold_ok and GRADS.sumsq is not None else None
        _lib.call("vlni_reduce_parts_sq", tab[0].data_ptr(), tab[1], tab[2], None if sq is None else sq.data_ptr(), SUMSQ_SLOTS, _st())
        if store is None and GRADS.arena is not None and STORE_PARTS and not torch.cuda.is_current_stream_capturing():
            # the same reduction as the next step will issue it if it has this step's structure (every destination stored, see GradArena.prebuild)
            flag = PART_STORE | (PART_SUMSQ if GRADS.sumsq is not None else 0)
            nxt = tuple((dst, part, n4, stride4, (eff & 0xffff) | (flag if GRADS.will_store(dst, 4 * n4) else 0))
                        for dst, part, n4, stride4, eff in entries if part)
            if nxt and nxt not in _PART_TABLES:
                _PART_TABLES[nxt] = _part_table(nxt, dev)
    if store is None and not _WQ and not torch.cuda.is_current_stream_capturing():
        GRADS.prebuild()


def _added(ws, bs):
    """An immediate (not queued) weight / bias gradient is about to be added into the arena (see GradArena.mixed)."""
    for prm in list(ws) + list(bs):
        prm._vlni_added = True
        if getattr(prm, "_vlni_queued", False):
            GRADS.mixed(prm.grad)


def _wb_grad_to(ws, bs, dy, x):
    """Weight + bias gradients of one (possibly row-packed) projection in a single pass over dy / x."""
    rows = [w.shape[0] for w in ws]
    if _direct(*ws, *bs, queue=True):
        wv = _packed_grad(ws) if len(ws) > 1 else ws[0].grad
        bv = _packed_grad(bs) if len(bs) > 1 else bs[0].grad
        if wv is not None and bv is not None:
            if getattr(ws[0], "_vlni_defer", False) and dy.dtype in H16 and dy.is_contiguous() and x.is_contiguous() \
                    and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0:
                ses = getattr(ws[0], "_vlni_auto", None)
                wq = ses.queue if (ses is not None and ses.active) else _WQ       # an agent model's open GradSession, or the trainer's queue
                ent = wq.get(wv.data_ptr())
                if ent is None:
                    ent = wq[wv.data_ptr()] = (wv, bv, [])
                    for prm in list(ws) + list(bs):
                        prm._vlni_queued = True   # its gradient may sit in a range the arena's zero fill skips (GradArena)
                        if getattr(prm, "_vlni_added", False):
                            GRADS.mixed(prm.grad)
                ent[2].append((dy, x))            # reduced later by flush_wgrads(): one grouped launch per parameter
            else:
                _added(ws, bs)
                wgrad(dy, x, out=wv, colsum_out=bv)
        else:
            _added(ws, bs)
            gw, gb = wgrad(dy, x, want_colsum=True)
            for prm, t in zip(ws, _split_rows(gw, rows)):
                prm.grad.add_(t)
            for prm, t in zip(bs, _split_rows(gb, rows)):
                prm.grad.add_(t)
        return (None,) * len(ws), (None,) * len(bs)
    gw, gb = wgrad(dy, x, want_colsum=True)
    return tuple(_split_rows(gw, rows)), tuple(_split_rows(gb, rows))


# ---- gradients of PLAIN parameters (no FlatTrainer): what an unchanged reference agent gets ---------------------------------------------------
# The agents call loss.backward() once per rollout (r2r/agent_cmt.py:827, map_nav_src/r2r/agent_base.py:223) and step their own optimizer.
# Through plain autograd every projection's weight gradient was one launch per STEP plus zero fills and AccumulateGrad adds (396 weight-
# gradient launches, 104 column sums per HAMT iteration), every LayerNorm's dgamma / dbeta a pair of zero-filled temporaries plus two adds -
# all host-bound. A model the wrappers marked (mark_agent_model) gets what train.FlatTrainer's gradient arena gives, for one backward pass at
# a time: the first operator of a backward pass that meets a marked parameter opens a GradSession - parameters without a gradient get views of
# one persistent, zero-filled float32 buffer as `.grad` (Q | K | V neighbours, so packed projections accumulate through one view) and the
# trainer's marks, so every kernel accumulates straight into them and the projections queue their (dY, X) pairs - and a callback the autograd
# engine runs at the END of the pass computes the queued weight gradients over all steps in grouped launches (flush_wgrads), gives parameters
# nothing touched their `None` back (an optimizer skips them, as it does in the reference) and takes the marks off again.
# What that changes for a caller: tensor hooks on these parameters and DDP's reducer do not see the gradients (the wrappers leave the mark off
# when torch.distributed runs more than one rank), torch.autograd.grad() does not return them, and `.grad` tensors are views of a buffer that
# is reused from one iteration to the next (as with optimizer.zero_grad(set_to_none=False)).
AUTO_DEFER = os.environ.get("VLNI_AUTO_DEFER", "1") == "1"


class GradSession:
    def __init__(self, module):
        units, seen = [], set()
        for mod in module.modules():
            if all(hasattr(mod, n) for n in ("query", "key", "value")):
                for attr in ("weight", "bias"):
                    unit = [getattr(getattr(mod, n), attr, None) for n in ("query", "key", "value")]
                    if all(p is not None and id(p) not in seen for p in unit):
                        units.append(unit)
                        seen.update(id(p) for p in unit)
        units += [[p] for p in module.parameters() if id(p) not in seen]
        self.params, self.off, n = [], {}, 0
        for unit in units:
            for p in unit:
                self.params.append(p)
                self.off[id(p)] = n
                n += (p.numel() + 7) // 8 * 8
        self.n, self.flat, self.queue = n, None, {}
        self.active, self.epoch, self.assigned, self.task = False, 0, [], -1
        # vln_imagine_amd/graphed.py: `hold` keeps the session open across the nested autograd pass of a backward-graph capture, `recorder`
        # collects the parameters that pass touches, `on_end` callbacks run once at the end of the agent's backward pass
        self.hold, self.recorder, self.on_end = False, None, []
        ref = weakref.ref(self)
        for p in self.params:
            p._vlni_auto = self

            def touched(prm, ref=ref):               # autograd itself accumulated into this parameter (an operator outside this library)
                ses = ref()
                if ses is not None:
                    prm._vlni_touch = ses.epoch
            if p.requires_grad:                      # (frozen parameters - the shipped fix_* configurations - never accumulate)
                p.register_post_accumulate_grad_hook(touched)

    def begin(self, register=True):
        """First marked parameter met in a backward pass: gradients to accumulate into, the trainer's marks, the end-of-pass callback.
        register=False (graphed.py, capturing a backward graph outside any backward pass): the caller closes the session with end(quiet=True)."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.epoch += 1
        self.assigned = [p for p in self.params if p.grad is None and p.requires_grad and p.dtype == torch.float32 and p.device == dev]
        # One fill of the whole buffer only when NO parameter still holds a view of it from an earlier pass (a second backward() without
        # zero_grad - gradient accumulation, two losses back-propagated one after the other - keeps the first pass's gradients there;
        # ADVICE round 5). Otherwise only the slots handed out now are zeroed.
        base = self.flat.untyped_storage().data_ptr()
        live = any(p.grad is not None and p.grad.is_cuda and p.grad.untyped_storage().data_ptr() == base for p in self.params)
        with torch.no_grad():
            if not live and len(self.assigned) > len(self.params) // 2:
                self.flat.zero_()
                for p in self.assigned:
                    o = self.off[id(p)]
                    p.grad = self.flat[o:o + p.numel()].view(p.shape)
            else:
                for p in self.assigned:
                    o = self.off[id(p)]
                    p.grad = self.flat[o:o + p.numel()].view(p.shape).zero_()
        for p in self.params:
            ok = p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
            p._vlni_direct = p._vlni_defer = ok
        self.active, self.task = True, torch._C._current_graph_task_id()
        if register:
            torch.autograd.Variable._execution_engine.queue_callback(self.end)

    def end(self, quiet=False):
        """End of the backward pass: the queued weight gradients over all their calls, `None` back where nothing arrived, marks off.
        quiet=True: a session opened only to CAPTURE a backward graph - nothing ran, every gradient handed out goes back to None."""
        try:
            with torch.no_grad():
                if self.queue and not quiet:
                    flush_wgrads(queue=self.queue, store=set())
        finally:
            self.queue.clear()
            ep = self.epoch
            for p in self.assigned:
                if quiet or getattr(p, "_vlni_touch", -1) != ep:
                    p.grad = None
            for p in self.params:
                p._vlni_direct = p._vlni_defer = False
            self.assigned, self.active = [], False
            cbs, self.on_end = self.on_end, []
            for cb in cbs:
                cb(self)


def enable_auto_defer(module, on=True):
    """Marks (or unmarks) a model's plain parameters for GradSession (see above)."""
    if on:
        have = {id(getattr(p, "_vlni_auto", None)) for p in module.parameters()}
        if len(have) == 1 and None not in {getattr(p, "_vlni_auto", None) for p in module.parameters()}:
            return next(iter(module.parameters()))._vlni_auto        # marked before (wrapper constructor + dropin.wrap_*)
        return GradSession(module)
    for p in module.parameters():
        p._vlni_auto = None
    return None


def mark_agent_model(module):
    """What the reference-facing wrappers (hamt.models.model_HAMT.VLNBertCMT, duet.models.model.VLNBert) do at construction: direct, grouped
    gradients for the model an agent will train with loss.backward() + its own optimizer - unless several ranks run (the agent then wraps
    the model in DistributedDataParallel, r2r/agent_cmt.py:61-63, whose reducer must see every gradient arrive through autograd)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    enable_auto_defer(module, on=not multi)
    return module


def _session(params):
    """The open GradSession of these parameters (opening it if this is a backward pass and they are marked), else None."""
    p0 = params[0]
    ses = getattr(p0, "_vlni_auto", None)
    if ses is None or not AUTO_DEFER:
        return None
    if ses.active and ses.task != torch._C._current_graph_task_id() and not ses.hold:
        ses.queue.clear()                  # a backward pass that raised never reached its callback: drop what it left behind
        ses.end()
    if not ses.active:
        if torch._C._current_graph_task_id() < 0 or torch.cuda.is_current_stream_capturing() or getattr(p0, "_vlni_direct", False):
            return None                    # not inside backward(); or a FlatTrainer owns these parameters
        ses.begin()
    ep = ses.epoch
    for p in params:
        if p is not None:
            p._vlni_touch = ep
    if ses.recorder is not None:
        ses.recorder.extend(params)
    return ses


def _ln_bwd_to(dy, x, g, b, mean, rstd, want, dres=None, drop=None):
    """LayerNorm backward with dgamma/dbeta accumulated in place when possible. With drop=(p, seed) a 4th value,
    dx * mask/(1-p), is returned (the gradient of the dropped dense output that fed this LayerNorm)."""
    if want and _direct(g, b):
        r = ln_bwd(dy, x, g, mean, rstd, g.grad, b.grad, dres=dres, drop=drop)
        return (r[0], None, None) + tuple(r[3:])
    return ln_bwd(dy, x, g, mean, rstd, want_param_grads=want, dres=dres, drop=drop)


# =====================================================================================
#  block-level calls: one Python -> C crossing per sublayer and direction (csrc/blocks.hip)
# =====================================================================================
# An eager caller - an unchanged reference agent on the drop-in modules - pays ~35-50 us of Python per kernel launch (argument marshalling,
# ctypes arrays, kernel-choice lookups): a dual self-attention block is 4 launches forward and 4 backward. The block entry points issue the
# same launches from C; Python allocates the activations (in the order the launch-by-launch code allocates them: an episode tape matches a
# recorded step's buffers to its ghost pass by sequence), fills one struct and crosses once. The launch-by-launch code below stays: it is the
# path of the first call of every shape (it times the GEMM pipelines and fills the kernel-choice cache the block calls read), of ghost passes,
# of float32 / unusual operands, and the reference the tests hold the block calls to (VLNI_BLOCK_CALLS=0).
BLOCK_CALLS = os.environ.get("VLNI_BLOCK_CALLS", "1") == "1"


class _BlkSide(ctypes.Structure):            # VlniBlockSide (include/vlni.h)
    _fields_ = [("B", ctypes.c_int), ("S", ctypes.c_int), ("x", ctypes.c_void_p), ("ldx", ctypes.c_long), ("kmask", ctypes.c_void_p),
                ("w_in", ctypes.c_void_p), ("b_in", ctypes.c_void_p), ("w_out", ctypes.c_void_p), ("b_out", ctypes.c_void_p),
                ("wt_in", ctypes.c_void_p), ("wt_out", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("seed_attn", ctypes.c_uint), ("seed_dense", ctypes.c_uint),
                ("mid", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("lse", ctypes.c_void_p), ("pre", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("dy", ctypes.c_void_p), ("lddy", ctypes.c_long), ("dpre", ctypes.c_void_p), ("dmid_drop", ctypes.c_void_p),
                ("daux", ctypes.c_void_p), ("dmid", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p)]


class _BlkArgs(ctypes.Structure):            # VlniBlockArgs
    _fields_ = [("dtype", ctypes.c_int), ("n", ctypes.c_int), ("H", ctypes.c_int), ("FF", ctypes.c_int), ("nh", ctypes.c_int),
                ("eps", ctypes.c_float), ("p_attn", ctypes.c_float), ("p_hidden", ctypes.c_float), ("act", ctypes.c_int), ("dact", ctypes.c_int),
                ("v_in", ctypes.c_int), ("v_out", ctypes.c_int), ("bias0", ctypes.c_void_p), ("dbias0", ctypes.c_void_p), ("s", _BlkSide * 2)]


# One zero-initialised argument struct PER CALL: forward runs on the caller's thread and Function.backward on the autograd
# engine's, so a module-global struct raced between a second model's forward and a backward pass, and kept the previous call's fields.


def _known_variant(dt, Ms, N, K, act, dact, res, pre):
    """The cached kernel choice of a (dual) GEMM launch exactly as gemm_nt / gemm_nt2 key it; None = not timed yet (take the launch-by-launch path,
    which times the pipelines)."""
    if len(Ms) == 2:
        if not AUTOTUNE:
            return 0
        key = (dt, Ms[0], Ms[1], N, K, act, dact, res, pre, False)
    else:
        if not AUTOTUNE or Ms[0] < 512:
            return 0
        key = (dt, Ms[0], N, K, act, dact, res, pre, False)
    v = _GEMM_BEST.get(key)
    if v is None and torch.cuda.is_current_stream_capturing():
        return 0
    return v


def _blk_ok(xs):
    return BLOCK_CALLS and not _ghost() and not NN_DGRAD and all(x.is_cuda for x in xs)


def _blk_seed(seed, elems, on):
    return _shift(seed, elems) if on else 0


def _blk_self_att_fwd(sides, bias0, eps, nh=12):
    """sides: [(x [B, S, H], kmask, drop, P)] (1 or 2). Returns per side (x2, qkv, c, lse, pre, y, mean, rstd), or None: take the launch-by-launch path."""
    n = len(sides)
    x0 = sides[0][0]
    dt, dev, H = x0.dtype, x0.device, x0.shape[2]
    Ms = [s[0].shape[0] * s[0].shape[1] for s in sides]
    v_in = _known_variant(dt, Ms, 3 * H, H, 0, 0, False, False)
    v_out = _known_variant(dt, Ms, H, H, 0, 0, True, False)
    if v_in is None or v_out is None or H != 64 * nh or any(s[0].dtype != dt or s[0].shape[2] != H for s in sides):
        return None               # (the C entry points score with 1 / sqrt(64): another head size takes the launch-by-launch path)
    pa = max(s[2][0] for s in sides)
    ph = max(s[2][1] for s in sides)
    x2 = [_rows(s[0]) for s in sides]
    mid = [_new((Ms[i], 3 * H), dt, dev) for i in range(n)]
    if n == 2 and dt in H16 and max(s[0].shape[1] for s in sides) <= 256:       # attn_fwd2's dual launch: contexts, then statistics
        aux = [_new((Ms[i], H), dt, dev) for i in range(n)]
        lse = [_new((sides[i][0].shape[0], nh, sides[i][0].shape[1]), torch.float32, dev) for i in range(n)]
    else:                                                                       # one attention launch per stream: (context, statistics) each
        aux, lse = [], []
        for i in range(n):
            aux.append(_new((Ms[i], H), dt, dev))
            lse.append(_new((sides[i][0].shape[0], nh, sides[i][0].shape[1]), torch.float32, dev))
    pre = [_new((Ms[i], H), dt, dev) for i in range(n)]
    y = [_new((Ms[i], H), dt, dev) for i in range(n)]
    mean = [_new((Ms[i],), torch.float32, dev) for i in range(n)]
    rstd = [_new((Ms[i],), torch.float32, dev) for i in range(n)]
    a = _BlkArgs()
    a.dtype, a.n, a.H, a.FF, a.nh, a.eps, a.p_attn, a.p_hidden = _DT[dt], n, H, 0, nh, eps, pa, ph
    a.v_in, a.v_out, a.bias0, a.dbias0 = v_in, v_out, _p(bias0) or None, None
    for i, (x, kmask, drop, P) in enumerate(sides):
        B, S = x.shape[0], x.shape[1]
        sd = a.s[i]
        sd.B, sd.S, sd.x, sd.ldx, sd.kmask = B, S, x2[i].data_ptr(), x2[i].stride(0), _p(kmask) or None
        sd.w_in, sd.b_in = _w((P[0], P[2], P[4]), dt).data_ptr(), _w((P[1], P[3], P[5]), torch.float32).data_ptr()
        sd.w_out, sd.b_out, sd.gamma, sd.beta = _w((P[6],), dt).data_ptr(), P[7].data_ptr(), P[8].data_ptr(), P[9].data_ptr()
        sd.seed_attn, sd.seed_dense = _blk_seed(drop[2], B * nh * S * S, pa > 0.0), _blk_seed(drop[2] + 1, Ms[i] * H, ph > 0.0)
        sd.mid, sd.aux, sd.lse, sd.pre = mid[i].data_ptr(), aux[i].data_ptr(), lse[i].data_ptr(), pre[i].data_ptr()
        sd.y, sd.mean, sd.rstd = y[i].data_ptr(), mean[i].data_ptr(), rstd[i].data_ptr()
    _lib.call("vlni_self_att_block_fwd", ctypes.addressof(a), _st())
    return [(x2[i], mid[i], aux[i], lse[i], pre[i], y[i], mean[i], rstd[i]) for i in range(n)]


def _blk_param_grads(want, g, b, dev, H):
    """(dgamma, dbeta, returned?) of one stream's LayerNorm: accumulated straight into the arena where the trainer marked the parameters."""
    if not want:
        return None, None, False
    if _direct(g, b):
        return g.grad, b.grad, False
    return torch.zeros((H,), dtype=torch.float32, device=dev), torch.zeros((H,), dtype=torch.float32, device=dev), True


def _blk_self_att_bwd(sides, bias0, want_dbias, nh=12):
    """sides: [(dy, x2, qkv, c, lse, pre, mean, rstd, kmask, drop, P, wants_params, wants_dx, B, S)]. Returns (per side (dx, grads dict), dbias0) or None.
    The weight / bias gradients go through _wb_grad_to exactly as in the launch-by-launch backward."""
    n = len(sides)
    dt, dev, H = sides[0][1].dtype, sides[0][1].device, sides[0][1].shape[1]
    Ms = [s[1].shape[0] for s in sides]
    nd = [i for i in range(n) if sides[i][12]]
    v_out = _known_variant(dt, Ms, H, H, 0, 0, False, False)
    v_in = _known_variant(dt, [Ms[i] for i in nd], H, 3 * H, 0, 0, True, False) if nd else 0
    if v_in is None or v_out is None or H != 64 * nh:
        return None
    pa = max(s[9][0] for s in sides)
    ph = max(s[9][1] for s in sides)
    a = _BlkArgs()
    a.dtype, a.n, a.H, a.FF, a.nh, a.eps, a.p_attn, a.p_hidden = _DT[dt], n, H, 0, nh, 0.0, pa, ph
    dbias0 = torch.zeros_like(bias0) if (bias0 is not None and want_dbias) else None
    a.v_in, a.v_out, a.bias0, a.dbias0 = v_in, v_out, _p(bias0) or None, _p(dbias0) or None
    keep, out = [], []
    for i, (dy, x2, qkv, c, lse, pre, mean, rstd, kmask, drop, P, wp, wdx, B, S) in enumerate(sides):
        dy2 = _rows(dy)
        dpre = torch.empty((Ms[i], H), dtype=dt, device=dev)
        dmd = torch.empty((Ms[i], H), dtype=dt, device=dev) if ph > 0.0 else dpre
        dc = torch.empty((Ms[i], H), dtype=dt, device=dev)
        dqkv = torch.empty_like(qkv)
        dx = torch.empty((Ms[i], H), dtype=dt, device=dev) if wdx else None
        dg, db, ret = _blk_param_grads(wp, P[8], P[9], dev, H)
        sd = a.s[i]
        sd.B, sd.S, sd.kmask = B, S, _p(kmask) or None
        sd.wt_in, sd.wt_out = _w((P[0], P[2], P[4]), dt, True).data_ptr(), _w((P[6],), dt, True).data_ptr()
        sd.gamma = P[8].data_ptr()
        sd.seed_attn, sd.seed_dense = drop[2] if pa > 0.0 else 0, (drop[2] + 1) if ph > 0.0 else 0
        sd.mid, sd.aux, sd.lse, sd.pre, sd.mean, sd.rstd = qkv.data_ptr(), c.data_ptr(), lse.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        sd.dy, sd.lddy, sd.dpre, sd.dmid_drop, sd.daux, sd.dmid = dy2.data_ptr(), dy2.stride(0), dpre.data_ptr(), dmd.data_ptr(), dc.data_ptr(), dqkv.data_ptr()
        sd.dx, sd.dgamma, sd.dbeta = _p(dx) or None, _p(dg) or None, _p(db) or None
        keep.append((dy2, dpre, dmd, dc, dqkv, dx, dg, db, ret))
    _lib.call("vlni_self_att_block_bwd", ctypes.addressof(a), _st())
    # weight gradients in the launch-by-launch backward's ORDER (out projections of all streams, then in projections): the deferred queue
    # keeps first-insertion order, and a captured flush must find the reduction table its warm-up step built
    g8s = [[None] * 8 for _ in range(n)]
    for i, sd_ in enumerate(sides):
        if sd_[11]:
            P = sd_[10]
            (g8s[i][6],), (g8s[i][7],) = _wb_grad_to((P[6],), (P[7],), keep[i][2], sd_[3])
    for i, sd_ in enumerate(sides):
        if sd_[11]:
            P = sd_[10]
            (g8s[i][0], g8s[i][2], g8s[i][4]), (g8s[i][1], g8s[i][3], g8s[i][5]) = _wb_grad_to((P[0], P[2], P[4]), (P[1], P[3], P[5]), keep[i][4], sd_[1])
    for i in range(n):
        dy2, dpre, dmd, dc, dqkv, dx, dg, db, ret = keep[i]
        out.append((dx, g8s[i], dg if ret else None, db if ret else None))
    return out, dbias0


def _blk_ffn_fwd(sides, eps):
    """sides: [(x, drop, P = (w1, b1, w2, b2, g, b))]. Returns per side (x2, z, h, pre, y, mean, rstd) or None."""
    n = len(sides)
    x0 = sides[0][0]
    dt, dev, H = x0.dtype, x0.device, x0.shape[-1]
    FF = sides[0][2][0].shape[0]
    x2 = [_rows(s[0]) for s in sides]
    Ms = [t.shape[0] for t in x2]
    act, dact = _gelu_codes(dt)
    v_in = _known_variant(dt, Ms, FF, H, act, 0, False, True)
    v_out = _known_variant(dt, Ms, H, FF, 0, 0, True, False)
    if v_in is None or v_out is None or any(s[0].dtype != dt or s[2][0].shape[0] != FF for s in sides):
        return None
    ph = max(s[1][1] for s in sides)
    z = [_new((Ms[i], FF), dt, dev) for i in range(n)]
    h = [_new((Ms[i], FF), dt, dev) for i in range(n)]
    pre = [_new((Ms[i], H), dt, dev) for i in range(n)]
    y = [_new((Ms[i], H), dt, dev) for i in range(n)]
    mean = [_new((Ms[i],), torch.float32, dev) for i in range(n)]
    rstd = [_new((Ms[i],), torch.float32, dev) for i in range(n)]
    a = _BlkArgs()
    a.dtype, a.n, a.H, a.FF, a.nh, a.eps, a.p_attn, a.p_hidden = _DT[dt], n, H, FF, 0, eps, 0.0, ph
    a.act, a.dact, a.v_in, a.v_out, a.bias0, a.dbias0 = act, dact, v_in, v_out, None, None
    for i, (x, drop, P) in enumerate(sides):
        sd = a.s[i]
        sd.B, sd.S, sd.x, sd.ldx, sd.kmask = Ms[i], 1, x2[i].data_ptr(), x2[i].stride(0), None
        sd.w_in, sd.b_in, sd.w_out, sd.b_out = _w((P[0],), dt).data_ptr(), P[1].data_ptr(), _w((P[2],), dt).data_ptr(), P[3].data_ptr()
        sd.gamma, sd.beta = P[4].data_ptr(), P[5].data_ptr()
        sd.seed_attn, sd.seed_dense = 0, _blk_seed(drop[2], Ms[i] * H, ph > 0.0)
        sd.mid, sd.aux, sd.lse, sd.pre = h[i].data_ptr(), z[i].data_ptr(), None, pre[i].data_ptr()
        sd.y, sd.mean, sd.rstd = y[i].data_ptr(), mean[i].data_ptr(), rstd[i].data_ptr()
    _lib.call("vlni_ffn_block_fwd", ctypes.addressof(a), _st())
    return [(x2[i], z[i], h[i], pre[i], y[i], mean[i], rstd[i]) for i in range(n)]


def _blk_ffn_bwd(sides):
    """sides: [(dy, x2, z, h, pre, mean, rstd, drop, P, wants_params, wants_dx)]. Returns per side (dx, g4, dgamma, dbeta) or None."""
    n = len(sides)
    dt, dev, H = sides[0][1].dtype, sides[0][1].device, sides[0][1].shape[1]
    FF = sides[0][2].shape[1]
    Ms = [s[1].shape[0] for s in sides]
    act, dact = _gelu_codes(dt)
    nd = [i for i in range(n) if sides[i][10]]
    v_out = _known_variant(dt, Ms, FF, H, 0, dact, False, False)
    v_in = _known_variant(dt, [Ms[i] for i in nd], H, FF, 0, 0, True, False) if nd else 0
    if v_in is None or v_out is None:
        return None
    ph = max(s[7][1] for s in sides)
    a = _BlkArgs()
    a.dtype, a.n, a.H, a.FF, a.nh, a.eps, a.p_attn, a.p_hidden = _DT[dt], n, H, FF, 0, 0.0, 0.0, ph
    a.act, a.dact, a.v_in, a.v_out, a.bias0, a.dbias0 = act, dact, v_in, v_out, None, None
    keep, out = [], []
    for i, (dy, x2, z, h, pre, mean, rstd, drop, P, wp, wdx) in enumerate(sides):
        dy2 = _rows(dy)
        dpre = torch.empty((Ms[i], H), dtype=dt, device=dev)
        dmd = torch.empty((Ms[i], H), dtype=dt, device=dev) if ph > 0.0 else dpre
        dz = torch.empty((Ms[i], FF), dtype=dt, device=dev)
        dx = torch.empty((Ms[i], H), dtype=dt, device=dev) if wdx else None
        dg, db, ret = _blk_param_grads(wp, P[4], P[5], dev, H)
        sd = a.s[i]
        sd.B, sd.S, sd.kmask = Ms[i], 1, None
        sd.wt_in, sd.wt_out, sd.gamma = _w((P[0],), dt, True).data_ptr(), _w((P[2],), dt, True).data_ptr(), P[4].data_ptr()
        sd.seed_attn, sd.seed_dense = 0, drop[2] if ph > 0.0 else 0
        sd.mid, sd.aux, sd.lse, sd.pre, sd.mean, sd.rstd = h.data_ptr(), z.data_ptr(), None, pre.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        sd.dy, sd.lddy, sd.dpre, sd.dmid_drop, sd.daux, sd.dmid = dy2.data_ptr(), dy2.stride(0), dpre.data_ptr(), dmd.data_ptr(), dz.data_ptr(), None
        sd.dx, sd.dgamma, sd.dbeta = _p(dx) or None, _p(dg) or None, _p(db) or None
        keep.append((dy2, dpre, dmd, dz, dx, dg, db, ret))
    _lib.call("vlni_ffn_block_bwd", ctypes.addressof(a), _st())
    g4s = [[None] * 4 for _ in range(n)]                     # (same order as the launch-by-launch backward: see _blk_self_att_bwd)
    for i, sd_ in enumerate(sides):
        if sd_[9]:
            P = sd_[8]
            (g4s[i][2],), (g4s[i][3],) = _wb_grad_to((P[2],), (P[3],), keep[i][2], sd_[3])
    for i, sd_ in enumerate(sides):
        if sd_[9]:
            P = sd_[8]
            (g4s[i][0],), (g4s[i][1],) = _wb_grad_to((P[0],), (P[1],), keep[i][3], sd_[1])
    for i in range(n):
        dy2, dpre, dmd, dz, dx, dg, db, ret = keep[i]
        out.append((dx, g4s[i], dg if ret else None, db if ret else None))
    return out


# =====================================================================================
#  sublayer autograd functions
# =====================================================================================
class _SelfAttBlock(torch.autograd.Function):
    """y = LN(dense(attn(x Wq, x Wk, x Wv)) + x): BertAttention, vilmodel_cmt.py:151-161."""

    @staticmethod
    def forward(ctx, x, kmask, bias, eps, drop, wq, bq, wk, bk, wv, bv, wo, bo, g, b):
        B, S, H = x.shape
        pa, ph, sd = drop
        if _blk_ok((x,)):
            r = _blk_self_att_fwd([(x, kmask, drop, (wq, bq, wk, bk, wv, bv, wo, bo, g, b))], bias, eps)
            if r is not None:
                x2, qkv, c, lse, pre, y, mean, rstd = r[0]
                ctx.save_for_backward(x2, qkv, c, lse, pre, mean, rstd, kmask, bias)
                ctx.P = (wq, bq, wk, bk, wv, bv, wo, bo, g, b)
                ctx.dims, ctx.drop = (B, S, H), drop
                return y.view(B, S, H)
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        wqkv, bqkv = _w((wq, wk, wv), dt), _w((bq, bk, bv), torch.float32)
        qkv = gemm_nt(x2, wqkv, bias=bqkv)
        c, lse = attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, S, kmask, bias, drop=(pa, sd))
        pre = gemm_nt(c, _w((wo,), dt), bias=bo, residual=x2, drop=(ph, sd + 1))
        y, mean, rstd = ln_fwd(pre, g, b, eps)
        ctx.save_for_backward(x2, qkv, c, lse, pre, mean, rstd, kmask, bias)
        ctx.P = (wq, bq, wk, bk, wv, bv, wo, bo, g, b)
        ctx.dims, ctx.drop = (B, S, H), drop
        return y.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        x2, qkv, c, lse, pre, mean, rstd, kmask, bias = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.P
        B, S, H = ctx.dims
        pa, ph, sd = ctx.drop
        dt = x2.dtype
        ng = ctx.needs_input_grad
        wparams = any(ng[5:])
        if _blk_ok((dy,)):
            r = _blk_self_att_bwd([(dy, x2, qkv, c, lse, pre, mean, rstd, kmask, ctx.drop, ctx.P, wparams, ng[0], B, S)], bias, bias is not None and ng[2])
            if r is not None:
                (dx, g8, dg, db), dbias = r[0][0], r[1]
                return (dx.view(B, S, H) if dx is not None else None, None, dbias, None, None) + tuple(g8) + (dg, db)
        dpre, dg, db, dpm = _ln_bwd_to(_rows(dy), pre, g, b, mean, rstd, wparams, drop=(ph, sd + 1))
        dwo = dbo = dwq = dwk = dwv = dbq = dbk = dbv = None
        if wparams:
            (dwo,), (dbo,) = _wb_grad_to((wo,), (bo,), dpm, c)
        dc = gemm_nt(dpm, _w((wo,), dt, True))
        dqkv = torch.empty_like(qkv)
        dbias = torch.zeros_like(bias) if (bias is not None and ng[2]) else None
        attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], c, dc, lse, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                 B, S, S, kmask, bias, dbias, drop=(pa, sd))
        if wparams:
            (dwq, dwk, dwv), (dbq, dbk, dbv) = _wb_grad_to((wq, wk, wv), (bq, bk, bv), dqkv, x2)
        dx = gemm_nt(dqkv, _w((wq, wk, wv), dt, True), residual=dpre).view(B, S, H) if ng[0] else None
        return dx, None, dbias, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg, db


class _FfnBlock(torch.autograd.Function):
    """y = LN(W2 gelu(W1 x + b1) + b2 + x): BertIntermediate + BertOutput, vilmodel_cmt.py:164-190."""

    @staticmethod
    def forward(ctx, x, eps, drop, w1, b1, w2, b2, g, b):
        shp = x.shape
        if _blk_ok((x,)):
            r = _blk_ffn_fwd([(x, drop, (w1, b1, w2, b2, g, b))], eps)
            if r is not None:
                x2, z, a, pre, y, mean, rstd = r[0]
                ctx.save_for_backward(x2, z, a, pre, mean, rstd)
                ctx.P = (w1, b1, w2, b2, g, b)
                ctx.shp, ctx.drop = shp, drop
                return y.view(shp)
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        z = _new((x2.shape[0], w1.shape[0]), dt, x.device)           # z, or GELU'(z) on the 16-bit paths (_gelu_codes)
        a = gemm_nt(x2, _w((w1,), dt), bias=b1, act=_gelu_codes(dt)[0], preact=z)
        pre = gemm_nt(a, _w((w2,), dt), bias=b2, residual=x2, drop=(drop[1], drop[2]))
        y, mean, rstd = ln_fwd(pre, g, b, eps)
        ctx.save_for_backward(x2, z, a, pre, mean, rstd)
        ctx.P = (w1, b1, w2, b2, g, b)
        ctx.shp, ctx.drop = shp, drop
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, z, a, pre, mean, rstd = ctx.saved_tensors
        w1, b1, w2, b2, g, b = ctx.P
        dt = x2.dtype
        ng = ctx.needs_input_grad
        wparams = any(ng[3:])
        if _blk_ok((dy,)):
            r = _blk_ffn_bwd([(dy, x2, z, a, pre, mean, rstd, ctx.drop, ctx.P, wparams, ng[0])])
            if r is not None:
                dx, g4, dg, db = r[0]
                return (dx.view(ctx.shp) if dx is not None else None, None, None) + tuple(g4) + (dg, db)
        dpre, dg, db, dpm = _ln_bwd_to(_rows(dy), pre, g, b, mean, rstd, wparams, drop=(ctx.drop[1], ctx.drop[2]))
        dw1 = db1 = dw2 = db2 = None
        if wparams:
            (dw2,), (db2,) = _wb_grad_to((w2,), (b2,), dpm, a)
        dz = gemm_nt(dpm, _w((w2,), dt, True), dact_src=z, dact=_gelu_codes(dt)[1])       # GELU' fused in the dgrad epilogue
        if wparams:
            (dw1,), (db1,) = _wb_grad_to((w1,), (b1,), dz, x2)
        dx = gemm_nt(dz, _w((w1,), dt, True), residual=dpre).view(ctx.shp) if ng[0] else None
        return dx, None, None, dw1, db1, dw2, db2, dg, db


class _XAttPairBlock(torch.autograd.Function):
    """Bidirectional cross-attention with ONE shared BertXAttention on the PRE-update inputs
    (LXRTXLayer.cross_att, vilmodel_cmt.py:385-397):
        lang' = LN(dense(attn(q=lang, kv=visn, mask_v)) + lang);  visn' = LN(dense(attn(q=visn, kv=lang, mask_l)) + visn)
    Q/K/V of both streams come from one packed [2304,768] projection each."""

    @staticmethod
    def forward(ctx, lang, visn, mask_l, mask_v, eps, drop, wq, bq, wk, bk, wv, bv, wo, bo, g, b):
        B, Sl, H = lang.shape
        pa, ph, sd = drop
        Sv = visn.shape[1]
        l2, v2 = _rows(_chk(lang, "lang")), _rows(_chk(visn, "visn"))
        dt = lang.dtype
        wqkv, bqkv, wo_c = _w((wq, wk, wv), dt), _w((bq, bk, bv), torch.float32), _w((wo,), dt)
        ql, qv = gemm_nt2((l2, v2), (wqkv, wqkv), bias=(bqkv, bqkv))
        (cl, lse_l), (cv, lse_v) = attn_fwd2((ql[:, :H], qv[:, :H]), (qv[:, H:2 * H], ql[:, H:2 * H]), (qv[:, 2 * H:], ql[:, 2 * H:]), B,
                                             (Sl, Sv), (Sv, Sl), (mask_v, mask_l), drop=(pa, (sd, sd + 1)))
        pre_l, pre_v = gemm_nt2((cl, cv), (wo_c, wo_c), bias=(bo, bo), residual=(l2, v2), drop=(ph, (sd + 2, sd + 3)))
        (yl, mean_l, rstd_l), (yv, mean_v, rstd_v) = ln_fwd2((pre_l, pre_v), (g, g), (b, b), eps)
        ctx.save_for_backward(l2, v2, ql, qv, cl, cv, lse_l, lse_v, pre_l, pre_v, mean_l, rstd_l, mean_v, rstd_v,
                              mask_l, mask_v)
        ctx.P = (wq, bq, wk, bk, wv, bv, wo, bo, g, b)
        ctx.dims, ctx.drop = (B, Sl, Sv, H), drop
        return yl.view(B, Sl, H), yv.view(B, Sv, H)

    @staticmethod
    def backward(ctx, dyl, dyv):
        (l2, v2, ql, qv, cl, cv, lse_l, lse_v, pre_l, pre_v, mean_l, rstd_l, mean_v, rstd_v,
         mask_l, mask_v) = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.P
        B, Sl, Sv, H = ctx.dims
        dt = l2.dtype
        ng = ctx.needs_input_grad
        pa, ph, sd = ctx.drop
        wparams = any(ng[6:])
        # capability check only for the projections (their gradients are QUEUED, not added to here: ADVICE round 5 - without queue=True the
        # check moved every cross-attention weight out of the arena's stored ranges); the LayerNorm vectors are added to by the kernel
        direct = wparams and _direct(wq, bq, wk, bk, wv, bv, wo, bo, queue=True) and _direct(g, b)
        dwo = dbo = dwq = dwk = dwv = dbq = dbk = dbv = dg = db = None
        if direct or not wparams:
            (dpl, _, _, dml), (dpv, _, _, dmv) = _ln_bwd_to2((_rows(dyl), _rows(dyv)), (pre_l, pre_v), (g, g), (b, b), (mean_l, mean_v),
                                                             (rstd_l, rstd_v), (wparams, wparams), drop=(ph, (sd + 2, sd + 3)))
        else:
            dpl, dg, db, dml = ln_bwd(_rows(dyl), pre_l, g, mean_l, rstd_l, drop=(ph, sd + 2))
            dpv, dg, db, dmv = ln_bwd(_rows(dyv), pre_v, g, mean_v, rstd_v, dg, db, drop=(ph, sd + 3))
        if direct:
            _wb_grad_to((wo,), (bo,), dml, cl); _wb_grad_to((wo,), (bo,), dmv, cv)
        elif wparams:
            dwo = wgrad(dmv, cv, wgrad(dml, cl))
            dbo = colsum(dmv, colsum(dml))
        wot = _w((wo,), dt, True)
        dcl, dcv = gemm_nt2((dml, dmv), (wot, wot))
        dql, dqv = torch.empty_like(ql), torch.empty_like(qv)
        attn_bwd2((ql[:, :H], qv[:, :H]), (qv[:, H:2 * H], ql[:, H:2 * H]), (qv[:, 2 * H:], ql[:, 2 * H:]), (cl, cv), (dcl, dcv),
                  (lse_l, lse_v), (dql[:, :H], dqv[:, :H]), (dqv[:, H:2 * H], dql[:, H:2 * H]), (dqv[:, 2 * H:], dql[:, 2 * H:]),
                  B, (Sl, Sv), (Sv, Sl), (mask_v, mask_l), drop=(pa, (sd, sd + 1)))
        if direct:
            _wb_grad_to((wq, wk, wv), (bq, bk, bv), dql, l2); _wb_grad_to((wq, wk, wv), (bq, bk, bv), dqv, v2)
        elif wparams:
            dwq, dwk, dwv = _split_rows(wgrad(dqv, v2, wgrad(dql, l2)), (H, H, H))
            dbq, dbk, dbv = _split_rows(colsum(dqv, colsum(dql)), (H, H, H))
        wt = _w((wq, wk, wv), dt, True)
        dl, dv = gemm_nt2((dql, dqv), (wt, wt), residual=(dpl, dpv))
        dl, dv = dl.view(B, Sl, H), dv.view(B, Sv, H)
        return dl, dv, None, None, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg, db


class _QkvProj(torch.autograd.Function):
    """qkv = x [Wq;Wk;Wv]^T + [bq;bk;bv] as its own node: the language stream that enters the FIRST cross-modal layer is the same at every step
    of an episode (r2r/agent_cmt.py:498-606 passes the same txt_embeds / imagine_embeds), and LXRTXLayer.cross_att projects the PRE-update
    inputs with shared weights (vilmodel_cmt.py:385-397) - so its Q / K / V are projected once per episode, autograd sums their gradient over
    the steps, and dgrad / wgrad of that projection run once on B x Sl rows instead of T times (HAMT side)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv):
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        qkv = gemm_nt(x2, _w((wq, wk, wv), dt), bias=_w((bq, bk, bv), torch.float32))
        ctx.save_for_backward(x2)
        ctx.P, ctx.shp = (wq, bq, wk, bk, wv, bv), x.shape
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        (x2,) = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv = ctx.P
        ng = ctx.needs_input_grad
        dqkv = dqkv.contiguous()
        dwq = dwk = dwv = dbq = dbk = dbv = None
        if any(ng[1:]):
            (dwq, dwk, dwv), (dbq, dbk, dbv) = _wb_grad_to((wq, wk, wv), (bq, bk, bv), dqkv, x2)
        dx = gemm_nt(dqkv, _w((wq, wk, wv), x2.dtype, True)).view(ctx.shp) if ng[0] else None
        return dx, dwq, dbq, dwk, dbk, dwv, dbv


class _XAttPairGivenQBlock(torch.autograd.Function):
    """_XAttPairBlock with the language stream's packed Q / K / V handed in (ops.qkv_proj of the same `lang`): only the vision stream is
    projected here. Gradients: d ql goes back to the projection node (summed over the episode's steps by autograd), d lang is the residual path
    only, the packed weight's gradient gets the vision rows here and the language rows in _QkvProj."""

    @staticmethod
    def forward(ctx, lang, ql, visn, mask_l, mask_v, eps, drop, wq, bq, wk, bk, wv, bv, wo, bo, g, b):
        B, Sl, H = lang.shape
        pa, ph, sd = drop
        Sv = visn.shape[1]
        l2, v2 = _rows(_chk(lang, "lang")), _rows(_chk(visn, "visn"))
        dt = lang.dtype
        assert ql.shape == (B * Sl, 3 * H) and ql.dtype == dt
        wqkv, bqkv, wo_c = _w((wq, wk, wv), dt), _w((bq, bk, bv), torch.float32), _w((wo,), dt)
        qv = gemm_nt(v2, wqkv, bias=bqkv)
        (cl, lse_l), (cv, lse_v) = attn_fwd2((ql[:, :H], qv[:, :H]), (qv[:, H:2 * H], ql[:, H:2 * H]), (qv[:, 2 * H:], ql[:, 2 * H:]), B,
                                             (Sl, Sv), (Sv, Sl), (mask_v, mask_l), drop=(pa, (sd, sd + 1)))
        pre_l, pre_v = gemm_nt2((cl, cv), (wo_c, wo_c), bias=(bo, bo), residual=(l2, v2), drop=(ph, (sd + 2, sd + 3)))
        (yl, mean_l, rstd_l), (yv, mean_v, rstd_v) = ln_fwd2((pre_l, pre_v), (g, g), (b, b), eps)
        ctx.save_for_backward(v2, ql, qv, cl, cv, lse_l, lse_v, pre_l, pre_v, mean_l, rstd_l, mean_v, rstd_v, mask_l, mask_v)
        ctx.P = (wq, bq, wk, bk, wv, bv, wo, bo, g, b)
        ctx.dims, ctx.drop = (B, Sl, Sv, H), drop
        return yl.view(B, Sl, H), yv.view(B, Sv, H)

    @staticmethod
    def backward(ctx, dyl, dyv):
        (v2, ql, qv, cl, cv, lse_l, lse_v, pre_l, pre_v, mean_l, rstd_l, mean_v, rstd_v, mask_l, mask_v) = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.P
        B, Sl, Sv, H = ctx.dims
        dt = v2.dtype
        ng = ctx.needs_input_grad
        pa, ph, sd = ctx.drop
        wparams = any(ng[7:])
        # capability check only for the projections (their gradients are QUEUED, not added to here: ADVICE round 5 - without queue=True the
        # check moved every cross-attention weight out of the arena's stored ranges); the LayerNorm vectors are added to by the kernel
        direct = wparams and _direct(wq, bq, wk, bk, wv, bv, wo, bo, queue=True) and _direct(g, b)
        dwo = dbo = dwq = dwk = dwv = dbq = dbk = dbv = dg = db = None
        if direct or not wparams:
            (dpl, _, _, dml), (dpv, _, _, dmv) = _ln_bwd_to2((_rows(dyl), _rows(dyv)), (pre_l, pre_v), (g, g), (b, b), (mean_l, mean_v),
                                                             (rstd_l, rstd_v), (wparams, wparams), drop=(ph, (sd + 2, sd + 3)))
        else:
            dpl, dg, db, dml = ln_bwd(_rows(dyl), pre_l, g, mean_l, rstd_l, drop=(ph, sd + 2))
            dpv, dg, db, dmv = ln_bwd(_rows(dyv), pre_v, g, mean_v, rstd_v, dg, db, drop=(ph, sd + 3))
        if direct:
            _wb_grad_to((wo,), (bo,), dml, cl); _wb_grad_to((wo,), (bo,), dmv, cv)
        elif wparams:
            dwo = wgrad(dmv, cv, wgrad(dml, cl))
            dbo = colsum(dmv, colsum(dml))
        wot = _w((wo,), dt, True)
        dcl, dcv = gemm_nt2((dml, dmv), (wot, wot))
        dql, dqv = torch.empty_like(ql), torch.empty_like(qv)
        attn_bwd2((ql[:, :H], qv[:, :H]), (qv[:, H:2 * H], ql[:, H:2 * H]), (qv[:, 2 * H:], ql[:, 2 * H:]), (cl, cv), (dcl, dcv),
                  (lse_l, lse_v), (dql[:, :H], dqv[:, :H]), (dqv[:, H:2 * H], dql[:, H:2 * H]), (dqv[:, 2 * H:], dql[:, 2 * H:]),
                  B, (Sl, Sv), (Sv, Sl), (mask_v, mask_l), drop=(pa, (sd, sd + 1)))
        if direct:
            _wb_grad_to((wq, wk, wv), (bq, bk, bv), dqv, v2)
        elif wparams:
            dwq, dwk, dwv = _split_rows(wgrad(dqv, v2), (H, H, H))
            dbq, dbk, dbv = _split_rows(colsum(dqv), (H, H, H))
        dv = gemm_nt(dqv, _w((wq, wk, wv), dt, True), residual=dpv).view(B, Sv, H) if ng[2] else None
        dl = dpl.view(B, Sl, H) if ng[0] else None
        return dl, (dql if ng[1] else None), dv, None, None, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg, db


class _DualSelfAttBlock(torch.autograd.Function):
    """Two independent BertAttention blocks (language and vision streams of an LXRT layer, vilmodel_cmt.py:399-407) whose
    projections run as dual-problem GEMM launches: y_i = LN_i(dense_i(attn(x_i Wq_i, x_i Wk_i, x_i Wv_i)) + x_i)."""

    @staticmethod
    def forward(ctx, x0, x1, km0, km1, bias0, eps, drop0, drop1, *P):
        P0, P1 = P[:10], P[10:]
        (B, S0, H), S1 = x0.shape, x1.shape[1]
        if _blk_ok((x0, x1)) and x0.dtype == x1.dtype and x1.shape[0] == B:
            r = _blk_self_att_fwd([(x0, km0, drop0, P0), (x1, km1, drop1, P1)], bias0, eps)
            if r is not None:
                (a0, q0, c0, lse0, pre0, y0, m0, r0), (a1, q1, c1, lse1, pre1, y1, m1, r1) = r
                ctx.save_for_backward(a0, a1, q0, q1, c0, c1, lse0, lse1, pre0, pre1, m0, r0, m1, r1, km0, km1, bias0)
                ctx.P, ctx.dims, ctx.drop = (P0, P1), (B, S0, S1, H), (drop0, drop1, max(drop0[1], drop1[1]))
                return y0.view(B, S0, H), y1.view(B, S1, H)
        a0, a1 = _rows(_chk(x0, "x0")), _rows(_chk(x1, "x1"))
        dt = x0.dtype
        wqkv = [_w((Pi[0], Pi[2], Pi[4]), dt) for Pi in (P0, P1)]
        bqkv = [_w((Pi[1], Pi[3], Pi[5]), torch.float32) for Pi in (P0, P1)]
        q0, q1 = gemm_nt2((a0, a1), wqkv, bias=bqkv)
        (c0, lse0), (c1, lse1) = attn_fwd2((q0[:, :H], q1[:, :H]), (q0[:, H:2 * H], q1[:, H:2 * H]), (q0[:, 2 * H:], q1[:, 2 * H:]), B,
                                           (S0, S1), (S0, S1), (km0, km1), bias0, drop=(max(drop0[0], drop1[0]), (drop0[2], drop1[2])))
        ph = max(drop0[1], drop1[1])
        pre0, pre1 = gemm_nt2((c0, c1), (_w((P0[6],), dt), _w((P1[6],), dt)), bias=(P0[7], P1[7]), residual=(a0, a1),
                              drop=(ph, (drop0[2] + 1, drop1[2] + 1)))
        (y0, m0, r0), (y1, m1, r1) = ln_fwd2((pre0, pre1), (P0[8], P1[8]), (P0[9], P1[9]), eps)
        ctx.save_for_backward(a0, a1, q0, q1, c0, c1, lse0, lse1, pre0, pre1, m0, r0, m1, r1, km0, km1, bias0)
        ctx.P, ctx.dims, ctx.drop = (P0, P1), (B, S0, S1, H), (drop0, drop1, ph)
        return y0.view(B, S0, H), y1.view(B, S1, H)

    @staticmethod
    def backward(ctx, dy0, dy1):
        a0, a1, q0, q1, c0, c1, lse0, lse1, pre0, pre1, m0, r0, m1, r1, km0, km1, bias0 = ctx.saved_tensors
        P0, P1 = ctx.P
        B, S0, S1, H = ctx.dims
        drop0, drop1, ph = ctx.drop
        dt = a0.dtype
        ng = ctx.needs_input_grad
        w0, w1 = any(ng[8:18]), any(ng[18:28])
        if _blk_ok((dy0, dy1)):
            r = _blk_self_att_bwd([(dy0, a0, q0, c0, lse0, pre0, m0, r0, km0, drop0, P0, w0, True, B, S0),
                                   (dy1, a1, q1, c1, lse1, pre1, m1, r1, km1, drop1, P1, w1, True, B, S1)], bias0, bias0 is not None and ng[4])
            if r is not None:
                ((dx0, g0, dg0, db0), (dx1, g1, dg1, db1)), dbias0 = r
                return (dx0.view(B, S0, H), dx1.view(B, S1, H), None, None, dbias0, None, None, None) + tuple(g0) + (dg0, db0) + tuple(g1) \
                    + (dg1, db1)
        dbias0 = torch.zeros_like(bias0) if (bias0 is not None and ng[4]) else None
        (dp0, dg0, db0, dm0), (dp1, dg1, db1, dm1) = _ln_bwd_to2((_rows(dy0), _rows(dy1)), (pre0, pre1), (P0[8], P1[8]), (P0[9], P1[9]),
                                                                 (m0, m1), (r0, r1), (w0, w1), drop=(ph, (drop0[2] + 1, drop1[2] + 1)))
        g0, g1 = [None] * 8, [None] * 8
        if w0:
            (g0[6],), (g0[7],) = _wb_grad_to((P0[6],), (P0[7],), dm0, c0)
        if w1:
            (g1[6],), (g1[7],) = _wb_grad_to((P1[6],), (P1[7],), dm1, c1)
        dc0, dc1 = gemm_nt2((dm0, dm1), (_w((P0[6],), dt, True), _w((P1[6],), dt, True)))
        dq0, dq1 = torch.empty_like(q0), torch.empty_like(q1)
        attn_bwd2((q0[:, :H], q1[:, :H]), (q0[:, H:2 * H], q1[:, H:2 * H]), (q0[:, 2 * H:], q1[:, 2 * H:]), (c0, c1), (dc0, dc1),
                  (lse0, lse1), (dq0[:, :H], dq1[:, :H]), (dq0[:, H:2 * H], dq1[:, H:2 * H]), (dq0[:, 2 * H:], dq1[:, 2 * H:]),
                  B, (S0, S1), (S0, S1), (km0, km1), bias0, dbias0, drop=(max(drop0[0], drop1[0]), (drop0[2], drop1[2])))
        if w0:
            (g0[0], g0[2], g0[4]), (g0[1], g0[3], g0[5]) = _wb_grad_to((P0[0], P0[2], P0[4]), (P0[1], P0[3], P0[5]), dq0, a0)
        if w1:
            (g1[0], g1[2], g1[4]), (g1[1], g1[3], g1[5]) = _wb_grad_to((P1[0], P1[2], P1[4]), (P1[1], P1[3], P1[5]), dq1, a1)
        dx0, dx1 = gemm_nt2((dq0, dq1), (_w((P0[0], P0[2], P0[4]), dt, True), _w((P1[0], P1[2], P1[4]), dt, True)),
                            residual=(dp0, dp1))
        return (dx0.view(B, S0, H), dx1.view(B, S1, H), None, None, dbias0, None, None, None) + tuple(g0) + (dg0, db0) + tuple(g1) \
            + (dg1, db1)


class _DualXAttQBlock(torch.autograd.Function):
    """Two independent cross-attention blocks against already projected contexts (DUET's global-map and local-viewpoint branches
    attend to the same text, vilmodel.py:384-399): y_i = LN_i(dense_i(attn(q = x_i Wq_i, kv_i)) + x_i), dual-problem GEMM launches."""

    @staticmethod
    def forward(ctx, x0, x1, kv0, kv1, mask_c, eps, drop0, drop1, *P):
        P0, P1 = P[:6], P[6:]                        # (wq, bq, wo, bo, g, b) per branch
        (B, S0, H), S1 = x0.shape, x1.shape[1]
        Sk = (kv0.shape[0] // B, kv1.shape[0] // B)             # the two contexts may differ (mask_c: one mask or a pair)
        mk0, mk1 = mask_c if isinstance(mask_c, (tuple, list)) else (mask_c, mask_c)
        a0, a1 = _rows(_chk(x0, "x0")), _rows(_chk(x1, "x1"))
        dt = x0.dtype
        q0, q1 = gemm_nt2((a0, a1), (_w((P0[0],), dt), _w((P1[0],), dt)), bias=(P0[1], P1[1]))
        (c0, lse0), (c1, lse1) = attn_fwd2((q0, q1), (kv0[:, :H], kv1[:, :H]), (kv0[:, H:], kv1[:, H:]), B, (S0, S1), Sk,
                                           (mk0, mk1), drop=(max(drop0[0], drop1[0]), (drop0[2], drop1[2])))
        ph = max(drop0[1], drop1[1])
        pre0, pre1 = gemm_nt2((c0, c1), (_w((P0[2],), dt), _w((P1[2],), dt)), bias=(P0[3], P1[3]), residual=(a0, a1),
                              drop=(ph, (drop0[2] + 1, drop1[2] + 1)))
        (y0, m0, r0), (y1, m1, r1) = ln_fwd2((pre0, pre1), (P0[4], P1[4]), (P0[5], P1[5]), eps)
        ctx.save_for_backward(a0, a1, q0, q1, kv0, kv1, c0, c1, lse0, lse1, pre0, pre1, m0, r0, m1, r1, mk0, mk1)
        ctx.P, ctx.dims, ctx.drop = (P0, P1), (B, S0, S1, Sk, H), (drop0, drop1, ph)
        return y0.view(B, S0, H), y1.view(B, S1, H)

    @staticmethod
    def backward(ctx, dy0, dy1):
        a0, a1, q0, q1, kv0, kv1, c0, c1, lse0, lse1, pre0, pre1, m0, r0, m1, r1, mk0, mk1 = ctx.saved_tensors
        P0, P1 = ctx.P
        B, S0, S1, Sk, H = ctx.dims
        drop0, drop1, ph = ctx.drop
        dt = a0.dtype
        ng = ctx.needs_input_grad
        w0, w1 = any(ng[8:14]), any(ng[14:20])
        (dp0, dg0, db0, dm0), (dp1, dg1, db1, dm1) = _ln_bwd_to2((_rows(dy0), _rows(dy1)), (pre0, pre1), (P0[4], P1[4]), (P0[5], P1[5]),
                                                                 (m0, m1), (r0, r1), (w0, w1), drop=(ph, (drop0[2] + 1, drop1[2] + 1)))
        g0, g1 = [None] * 4, [None] * 4
        if w0:
            (g0[2],), (g0[3],) = _wb_grad_to((P0[2],), (P0[3],), dm0, c0)
        if w1:
            (g1[2],), (g1[3],) = _wb_grad_to((P1[2],), (P1[3],), dm1, c1)
        dc0, dc1 = gemm_nt2((dm0, dm1), (_w((P0[2],), dt, True), _w((P1[2],), dt, True)))
        dq0, dq1 = torch.empty_like(q0), torch.empty_like(q1)
        dkv0, dkv1 = torch.empty_like(kv0), torch.empty_like(kv1)
        attn_bwd2((q0, q1), (kv0[:, :H], kv1[:, :H]), (kv0[:, H:], kv1[:, H:]), (c0, c1), (dc0, dc1), (lse0, lse1), (dq0, dq1),
                  (dkv0[:, :H], dkv1[:, :H]), (dkv0[:, H:], dkv1[:, H:]), B, (S0, S1), Sk, (mk0, mk1),
                  drop=(max(drop0[0], drop1[0]), (drop0[2], drop1[2])))
        if w0:
            (g0[0],), (g0[1],) = _wb_grad_to((P0[0],), (P0[1],), dq0, a0)
        if w1:
            (g1[0],), (g1[1],) = _wb_grad_to((P1[0],), (P1[1],), dq1, a1)
        dx0, dx1 = gemm_nt2((dq0, dq1), (_w((P0[0],), dt, True), _w((P1[0],), dt, True)), residual=(dp0, dp1))
        return (dx0.view(B, S0, H), dx1.view(B, S1, H), dkv0 if ng[2] else None, dkv1 if ng[3] else None, None, None, None, None) \
            + tuple(g0) + (dg0, db0) + tuple(g1) + (dg1, db1)


class _DualFfnBlock(torch.autograd.Function):
    """Two independent FFN blocks (vilmodel_cmt.py:409-421) with dual-problem GEMM launches."""

    @staticmethod
    def forward(ctx, x0, x1, eps, drop0, drop1, *P):
        P0, P1 = P[:6], P[6:]
        s0, s1 = x0.shape, x1.shape
        if _blk_ok((x0, x1)) and x0.dtype == x1.dtype:
            r = _blk_ffn_fwd([(x0, drop0, P0), (x1, drop1, P1)], eps)
            if r is not None:
                (a0, z0, h0, pre0, y0, m0, r0), (a1, z1, h1, pre1, y1, m1, r1) = r
                ctx.save_for_backward(a0, a1, z0, z1, h0, h1, pre0, pre1, m0, r0, m1, r1)
                ctx.P, ctx.shp, ctx.drop = (P0, P1), (s0, s1), (drop0, drop1, max(drop0[1], drop1[1]))
                return y0.view(s0), y1.view(s1)
        a0, a1 = _rows(_chk(x0, "x0")), _rows(_chk(x1, "x1"))
        dt = x0.dtype
        FF = P0[0].shape[0]
        z0 = _new((a0.shape[0], FF), dt, x0.device)
        z1 = _new((a1.shape[0], FF), dt, x0.device)
        h0, h1 = gemm_nt2((a0, a1), (_w((P0[0],), dt), _w((P1[0],), dt)), bias=(P0[1], P1[1]), act=_gelu_codes(dt)[0], preact=(z0, z1))
        ph = max(drop0[1], drop1[1])
        pre0, pre1 = gemm_nt2((h0, h1), (_w((P0[2],), dt), _w((P1[2],), dt)), bias=(P0[3], P1[3]), residual=(a0, a1),
                              drop=(ph, (drop0[2], drop1[2])))
        (y0, m0, r0), (y1, m1, r1) = ln_fwd2((pre0, pre1), (P0[4], P1[4]), (P0[5], P1[5]), eps)
        ctx.save_for_backward(a0, a1, z0, z1, h0, h1, pre0, pre1, m0, r0, m1, r1)
        ctx.P, ctx.shp, ctx.drop = (P0, P1), (s0, s1), (drop0, drop1, ph)
        return y0.view(s0), y1.view(s1)

    @staticmethod
    def backward(ctx, dy0, dy1):
        a0, a1, z0, z1, h0, h1, pre0, pre1, m0, r0, m1, r1 = ctx.saved_tensors
        P0, P1 = ctx.P
        drop0, drop1, ph = ctx.drop
        dt = a0.dtype
        ng = ctx.needs_input_grad
        w0, w1 = any(ng[5:11]), any(ng[11:17])
        if _blk_ok((dy0, dy1)):
            r = _blk_ffn_bwd([(dy0, a0, z0, h0, pre0, m0, r0, drop0, P0, w0, True), (dy1, a1, z1, h1, pre1, m1, r1, drop1, P1, w1, True)])
            if r is not None:
                (dx0, g0, dg0, db0), (dx1, g1, dg1, db1) = r
                return (dx0.view(ctx.shp[0]), dx1.view(ctx.shp[1]), None, None, None) + tuple(g0) + (dg0, db0) + tuple(g1) + (dg1, db1)
        (dp0, dg0, db0, dm0), (dp1, dg1, db1, dm1) = _ln_bwd_to2((_rows(dy0), _rows(dy1)), (pre0, pre1), (P0[4], P1[4]), (P0[5], P1[5]),
                                                                 (m0, m1), (r0, r1), (w0, w1), drop=(ph, (drop0[2], drop1[2])))
        g0, g1 = [None] * 4, [None] * 4
        if w0:
            (g0[2],), (g0[3],) = _wb_grad_to((P0[2],), (P0[3],), dm0, h0)
        if w1:
            (g1[2],), (g1[3],) = _wb_grad_to((P1[2],), (P1[3],), dm1, h1)
        dz0, dz1 = gemm_nt2((dm0, dm1), (_w((P0[2],), dt, True), _w((P1[2],), dt, True)), dact_src=(z0, z1), dact=_gelu_codes(dt)[1])
        if w0:
            (g0[0],), (g0[1],) = _wb_grad_to((P0[0],), (P0[1],), dz0, a0)
        if w1:
            (g1[0],), (g1[1],) = _wb_grad_to((P1[0],), (P1[1],), dz1, a1)
        dx0, dx1 = gemm_nt2((dz0, dz1), (_w((P0[0],), dt, True), _w((P1[0],), dt, True)), residual=(dp0, dp1))
        return (dx0.view(ctx.shp[0]), dx1.view(ctx.shp[1]), None, None, None) + tuple(g0) + (dg0, db0) + tuple(g1) + (dg1, db1)


class _XAttBlock(torch.autograd.Function):
    """One-directional cross-attention: y = LN(dense(attn(q=x, kv=ctx_in, mask)) + x)
    (GraphLXRTXLayer cross step, VLN-DUET vilmodel.py:384-399; HAMT no_lang_ca)."""

    @staticmethod
    def forward(ctx, x, c_in, mask_c, eps, drop, wq, bq, wk, bk, wv, bv, wo, bo, g, b):
        B, Sq, H = x.shape
        pa, ph, sd = drop
        Sk = c_in.shape[1]
        x2, c2 = _rows(_chk(x, "x")), _rows(_chk(c_in, "context"))
        dt = x.dtype
        q = gemm_nt(x2, _w((wq,), dt), bias=bq)
        kv = gemm_nt(c2, _w((wk, wv), dt), bias=_w((bk, bv), torch.float32))
        a, lse = attn_fwd(q, kv[:, :H], kv[:, H:], B, Sq, Sk, mask_c, drop=(pa, sd))
        pre = gemm_nt(a, _w((wo,), dt), bias=bo, residual=x2, drop=(ph, sd + 1))
        y, mean, rstd = ln_fwd(pre, g, b, eps)
        ctx.save_for_backward(x2, c2, q, kv, a, lse, pre, mean, rstd, mask_c)
        ctx.P = (wq, bq, wk, bk, wv, bv, wo, bo, g, b)
        ctx.dims, ctx.drop = (B, Sq, Sk, H), drop
        return y.view(B, Sq, H)

    @staticmethod
    def backward(ctx, dy):
        x2, c2, q, kv, a, lse, pre, mean, rstd, mask_c = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.P
        B, Sq, Sk, H = ctx.dims
        dt = x2.dtype
        ng = ctx.needs_input_grad
        pa, ph, sd = ctx.drop
        wparams = any(ng[5:])
        dpre, dg, db, dpm = _ln_bwd_to(_rows(dy), pre, g, b, mean, rstd, wparams, drop=(ph, sd + 1))
        dwo = dbo = dwq = dbq = dwk = dwv = dbk = dbv = None
        if wparams:
            (dwo,), (dbo,) = _wb_grad_to((wo,), (bo,), dpm, a)
        da = gemm_nt(dpm, _w((wo,), dt, True))
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        attn_bwd(q, kv[:, :H], kv[:, H:], a, da, lse, dq, dkv[:, :H], dkv[:, H:], B, Sq, Sk, mask_c, drop=(pa, sd))
        if wparams:
            (dwq,), (dbq,) = _wb_grad_to((wq,), (bq,), dq, x2)
            (dwk, dwv), (dbk, dbv) = _wb_grad_to((wk, wv), (bk, bv), dkv, c2)
        dx = gemm_nt(dq, _w((wq,), dt, True), residual=dpre).view(B, Sq, H) if ng[0] else None
        dc = gemm_nt(dkv, _w((wk, wv), dt, True)).view(B, Sk, H) if ng[1] else None
        return dx, dc, None, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg, db


class _KvProj(torch.autograd.Function):
    """kv = ctx_in [Wk;Wv]^T + [bk;bv]: the context side of a cross-attention, as its own node so that a context that
    does not change over the steps of an episode (DUET never updates the text stream, vilmodel.py:384-399) is projected ONCE
    and its gradient is reduced once."""

    @staticmethod
    def forward(ctx, c_in, wk, bk, wv, bv):
        c2 = _rows(_chk(c_in, "context"))
        kv = gemm_nt(c2, _w((wk, wv), c2.dtype), bias=_w((bk, bv), torch.float32))
        ctx.save_for_backward(c2)
        ctx.P, ctx.shp = (wk, bk, wv, bv), c_in.shape
        return kv

    @staticmethod
    def backward(ctx, dkv):
        (c2,) = ctx.saved_tensors
        wk, bk, wv, bv = ctx.P
        ng = ctx.needs_input_grad
        dkv = dkv.contiguous()
        dwk = dwv = dbk = dbv = None
        if any(ng[1:]):
            (dwk, dwv), (dbk, dbv) = _wb_grad_to((wk, wv), (bk, bv), dkv, c2)
        dc = gemm_nt(dkv, _w((wk, wv), c2.dtype, True)).view(ctx.shp) if ng[0] else None
        return dc, dwk, dbk, dwv, dbv


class _XAttQBlock(torch.autograd.Function):
    """y = LN(dense(attn(q = x Wq, k/v = given projected context)) + x): the query side of _XAttBlock."""

    @staticmethod
    def forward(ctx, x, kv, mask_c, eps, drop, wq, bq, wo, bo, g, b):
        B, Sq, H = x.shape
        Sk = kv.shape[0] // B
        pa, ph, sd = drop
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        q = gemm_nt(x2, _w((wq,), dt), bias=bq)
        a, lse = attn_fwd(q, kv[:, :H], kv[:, H:], B, Sq, Sk, mask_c, drop=(pa, sd))
        pre = gemm_nt(a, _w((wo,), dt), bias=bo, residual=x2, drop=(ph, sd + 1))
        y, mean, rstd = ln_fwd(pre, g, b, eps)
        ctx.save_for_backward(x2, q, kv, a, lse, pre, mean, rstd, mask_c)
        ctx.P, ctx.dims, ctx.drop = (wq, bq, wo, bo, g, b), (B, Sq, Sk, H), drop
        return y.view(B, Sq, H)

    @staticmethod
    def backward(ctx, dy):
        x2, q, kv, a, lse, pre, mean, rstd, mask_c = ctx.saved_tensors
        wq, bq, wo, bo, g, b = ctx.P
        B, Sq, Sk, H = ctx.dims
        pa, ph, sd = ctx.drop
        dt = x2.dtype
        ng = ctx.needs_input_grad
        wparams = any(ng[5:])
        dpre, dg, db, dpm = _ln_bwd_to(_rows(dy), pre, g, b, mean, rstd, wparams, drop=(ph, sd + 1))
        dwo = dbo = dwq = dbq = None
        if wparams:
            (dwo,), (dbo,) = _wb_grad_to((wo,), (bo,), dpm, a)
        da = gemm_nt(dpm, _w((wo,), dt, True))
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        attn_bwd(q, kv[:, :H], kv[:, H:], a, da, lse, dq, dkv[:, :H], dkv[:, H:], B, Sq, Sk, mask_c, drop=(pa, sd))
        if wparams:
            (dwq,), (dbq,) = _wb_grad_to((wq,), (bq,), dq, x2)
        dx = gemm_nt(dq, _w((wq,), dt, True), residual=dpre).view(B, Sq, H) if ng[0] else None
        return dx, (dkv if ng[1] else None), None, None, None, dwq, dbq, dwo, dbo, dg, db


class _PreNormAttBlock(torch.autograd.Function):
    """y = x + out_proj(attn(in_proj(LN(x)))) with a packed [2304,768] in_proj and a key-padding mask (-inf):
    TransformerEncoderLayer.forward_pre, first half (VLN-DUET models/transformer.py:170-177)."""

    @staticmethod
    def forward(ctx, x, kmask, eps, drop, g, b, win, bin_, wo, bo):
        B, S, H = x.shape
        pa, ph, sd = drop
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        xn, mean, rstd = ln_fwd(x2, g, b, eps)
        qkv = gemm_nt(xn, _w((win,), dt), bias=bin_)
        c, lse = attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, S, kmask, drop=(pa, sd))
        y = gemm_nt(c, _w((wo,), dt), bias=bo, residual=x2, drop=(ph, sd + 1))
        ctx.save_for_backward(x2, xn, qkv, c, lse, mean, rstd, kmask)
        ctx.P, ctx.dims, ctx.drop = (g, b, win, bin_, wo, bo), (B, S, H), drop
        return y.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        x2, xn, qkv, c, lse, mean, rstd, kmask = ctx.saved_tensors
        g, b, win, bin_, wo, bo = ctx.P
        B, S, H = ctx.dims
        dt = x2.dtype
        ng = ctx.needs_input_grad
        pa, ph, sd = ctx.drop
        wparams = any(ng[4:])
        dy2 = _rows(dy)
        dym = dropout_apply(dy2, ph, sd + 1)
        dwo = dbo = dwin = dbin = None
        if wparams:
            (dwo,), (dbo,) = _wb_grad_to((wo,), (bo,), dym, c)
        dc = gemm_nt(dym, _w((wo,), dt, True))
        dqkv = torch.empty_like(qkv)
        attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], c, dc, lse, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                 B, S, S, kmask, drop=(pa, sd))
        if wparams:
            (dwin,), (dbin,) = _wb_grad_to((win,), (bin_,), dqkv, xn)
        dxn = gemm_nt(dqkv, _w((win,), dt, True))
        dx, dg, db = _ln_bwd_to(dxn, x2, g, b, mean, rstd, wparams, dres=dy2)
        return dx.view(B, S, H), None, None, None, dg, db, dwin, dbin, dwo, dbo


class _PreNormFfnBlock(torch.autograd.Function):
    """y = x + W2 gelu(W1 LN(x) + b1) + b2: forward_pre, second half (transformer.py:178-181)."""

    @staticmethod
    def forward(ctx, x, eps, drop, g, b, w1, b1, w2, b2):
        shp = x.shape
        x2 = _rows(_chk(x, "x"))
        dt = x.dtype
        _, ph, sd = drop
        xn, mean, rstd = ln_fwd(x2, g, b, eps)
        z = _new((x2.shape[0], w1.shape[0]), dt, x.device)
        a = gemm_nt(xn, _w((w1,), dt), bias=b1, act=_gelu_codes(dt)[0], preact=z, drop=(ph, sd))
        y = gemm_nt(a, _w((w2,), dt), bias=b2, residual=x2, drop=(ph, sd + 1))
        ctx.save_for_backward(x2, xn, z, a, mean, rstd)
        ctx.P, ctx.shp, ctx.drop = (g, b, w1, b1, w2, b2), shp, drop
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, xn, z, a, mean, rstd = ctx.saved_tensors
        g, b, w1, b1, w2, b2 = ctx.P
        dt = x2.dtype
        ng = ctx.needs_input_grad
        _, ph, sd = ctx.drop
        wparams = any(ng[3:])
        dy2 = _rows(dy)
        dym = dropout_apply(dy2, ph, sd + 1)
        dw1 = db1 = dw2 = db2 = None
        if wparams:
            (dw2,), (db2,) = _wb_grad_to((w2,), (b2,), dym, a)
        dz = gemm_nt(dym, _w((w2,), dt, True), dact_src=z, dact=_gelu_codes(dt)[1], drop=(ph, sd))
        if wparams:
            (dw1,), (db1,) = _wb_grad_to((w1,), (b1,), dz, xn)
        dxn = gemm_nt(dz, _w((w1,), dt, True))
        dx, dg, db = _ln_bwd_to(dxn, x2, g, b, mean, rstd, wparams, dres=dy2)
        return dx.view(ctx.shp), None, None, dg, db, dw1, db1, dw2, db2


# =====================================================================================
#  generic operators
# =====================================================================================
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual); act 0 none / 1 gelu / 2 relu."""

    @staticmethod
    def forward(ctx, x, w, b, act, out_dtype):
        shp = x.shape
        dt = out_dtype
        x2 = _rows(_chk(x, "x"))
        if x2.dtype != dt:
            x2 = cast(x2, dt, tape=True)
        z = _new((x2.shape[0], w.shape[0]), dt, x.device) if act else None
        y = gemm_nt(x2, _w((w,), dt), bias=b, act=act, preact=z)
        ctx.save_for_backward(x2, z, w)
        ctx.act, ctx.shp, ctx.in_dtype, ctx.b = act, shp, x.dtype, b
        return y.view(shp[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, z, w = ctx.saved_tensors
        dt = x2.dtype
        ng = ctx.needs_input_grad
        dy2 = _rows(dy)
        if ctx.act:
            dz = torch.empty_like(dy2)
            _lib.call("vlni_act_bwd", _dt(dy2), ctx.act, dy2.data_ptr(), z.data_ptr(), dz.data_ptr(), dy2.numel(), _st())
            dy2 = dz
        dw = db = None
        if ng[1] and ng[2]:
            (dw,), (db,) = _wb_grad_to((w,), (ctx.b,), dy2, x2)
        elif ng[1]:
            dw = _wgrad_to((w,), dy2, x2)[0]
        elif ng[2]:
            db = _bgrad_to((ctx.b,), dy2)[0]
        dx = None
        if ng[0]:
            dx = gemm_nt(dy2, _w((w,), dt, True))
            if dx.dtype != ctx.in_dtype:
                dx = cast(dx, ctx.in_dtype)
            dx = dx.view(ctx.shp)
        return dx, dw, db, None, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        shp = x.shape
        x2 = _rows(_chk(x, "x"))
        y, mean, rstd = ln_fwd(x2, g, b, eps)
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.shp, ctx.b = shp, b
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        wp = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dg, db = _ln_bwd_to(_rows(dy), x2, g, ctx.b, mean, rstd, wp)
        return dx.view(ctx.shp), dg, db, None


class _SmallKLinear(torch.autograd.Function):
    """y = x W^T + b for K <= 16 float32 features (angle / position features)."""

    @staticmethod
    def forward(ctx, x, w, b, out_dtype):
        shp = x.shape
        x2 = _rows(_chk(x, "x")).float()
        N, K = w.shape
        y = _new((x2.shape[0], N), out_dtype, x.device)
        if not _ghost():
            _lib.call("vlni_smallk_linear_fwd", _DT[out_dtype], x2.data_ptr(), x2.stride(0), w.data_ptr(), _p(b), y.data_ptr(),
                      y.stride(0), x2.shape[0], N, K, _st())
        ctx.save_for_backward(x2, w)
        ctx.has_b, ctx.b = b is not None, b
        return y.view(shp[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        N, K = w.shape
        dy2 = _rows(dy)
        b = ctx.b
        if _direct(w) and (b is None or _direct(b)):         # accumulate straight into the gradient arena
            _lib.call("vlni_smallk_linear_bwd", _dt(dy2), dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0),
                      w.grad.data_ptr(), _p(b.grad if b is not None else None), x2.shape[0], N, K, _st())
            return None, None, None, None
        dw = torch.zeros_like(w)
        db = torch.zeros((N,), dtype=torch.float32, device=w.device) if ctx.has_b else None
        _lib.call("vlni_smallk_linear_bwd", _dt(dy2), dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0),
                  dw.data_ptr(), _p(db), x2.shape[0], N, K, _st())
        return None, dw, db, None


class _SumLayerNorm(torch.autograd.Function):
    """y = LN(sum_k src_k). spec[k] = ('dense' | 'bcast' | 'gather', is_param)."""

    @staticmethod
    def forward(ctx, spec, idxs, eps, out_dtype, rows, g, b, *srcs):
        H = g.shape[0]
        n = len(srcs)
        dev = g.device
        ptrs = (ctypes.c_void_p * n)()
        lds = (ctypes.c_long * n)()
        ips = (ctypes.c_void_p * n)()
        f32 = (ctypes.c_int * n)()
        keep = []
        for k, (s, (kind, _)) in enumerate(zip(srcs, spec)):
            _chk(s, "source")
            s2 = s.reshape(-1, H)
            s2 = s2 if s2.stride(1) == 1 else s2.contiguous()
            keep.append(s2)
            ptrs[k] = s2.data_ptr()
            lds[k] = 0 if kind == "bcast" else s2.stride(0)
            ips[k] = idxs[k].data_ptr() if kind == "gather" else None
            f32[k] = 1 if s2.dtype == torch.float32 else 0
            if s2.dtype != torch.float32 and s2.dtype != out_dtype:
                raise TypeError("sum_layernorm: activation sources must have the compute dtype")
        y = _new((rows, H), out_dtype, dev)
        xs = _new((rows, H), out_dtype, dev)
        mean = _new((rows,), torch.float32, dev)
        rstd = _new((rows,), torch.float32, dev)
        if not _ghost():
            _lib.call("vlni_sum_layernorm_fwd", _DT[out_dtype], n, ptrs, lds, ips, f32, g.data_ptr(), b.data_ptr(), eps,
                      y.data_ptr(), H, xs.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), rows, H, _st())
        ctx.save_for_backward(xs, g, mean, rstd, *[i for i in idxs if i is not None])
        ctx.spec, ctx.idx_pos = spec, [k for k, i in enumerate(idxs) if i is not None]
        ctx.shapes = [(s.shape, s.dtype) for s in srcs]
        ctx.b, ctx.tables = b, [s if k == "gather" else None for s, (k, _) in zip(srcs, spec)]
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, g, mean, rstd, *idl = ctx.saved_tensors
        idxs = dict(zip(ctx.idx_pos, idl))
        ng = ctx.needs_input_grad
        dsum, dg, db = _ln_bwd_to(_rows(dy), xs, g, ctx.b, mean, rstd, ng[5] or ng[6])
        rows, H = dsum.shape
        grads = []
        for k, ((kind, _), (shape, dtype)) in enumerate(zip(ctx.spec, ctx.shapes)):
            if not ng[7 + k]:
                grads.append(None)
            elif kind == "dense":
                grads.append((dsum if dtype == dsum.dtype else cast(dsum, dtype)).view(shape))
            elif kind == "bcast":
                grads.append(colsum(dsum).view(shape).to(dtype))
            else:
                tab = ctx.tables[k]
                direct = _direct(tab) and tab.grad.is_contiguous()
                tg = tab.grad if direct else torch.zeros(shape, dtype=torch.float32, device=dsum.device)
                if SMALL_TABLE_SCATTER and shape[0] <= 8:      # a few table rows, thousands of adders each: block-level accumulators
                    _lib.call("vlni_scatter_add_rows_small", _dt(dsum), dsum.data_ptr(), dsum.stride(0), idxs[k].data_ptr(),
                              tg.data_ptr(), rows, H, shape[0], _st())
                else:
                    _lib.call("vlni_scatter_add_rows", _dt(dsum), dsum.data_ptr(), dsum.stride(0), idxs[k].data_ptr(),
                              tg.data_ptr(), rows, H, shape[0] if idxs[k] is not None else 1, _st())
                grads.append(None if direct else tg)
        return (None, None, None, None, None, dg, db) + tuple(grads)


class _EmbedCombine(torch.autograd.Function):
    """y = dropout([LN_o]([LN_a](a) + LN_b(f W_b^T + b_b) + row + table[idx] + table2[idx2] + extra)) - the observation / history / panorama /
    map-node embeddings in ONE launch (vlni_embed_combine_fwd; ImageEmbeddings vilmodel_cmt.py:535-544, HistoryEmbeddings :596-618, DUET
    vilmodel.py:1087-1131,1140-1156). The backward runs the operators the unfused graph ran (dropout mask, LayerNorm backward x <= 3, the
    small-K linear's weight gradient, column sum, row scatter): it is batched over the episode anyway. Optional parts are None."""

    @staticmethod
    def forward(ctx, a, f, extra, idx, idx2, eps, p_drop, seed, out_dtype, ga, ba, Wb, bb, gb, beb, row, table, table2, go, bo):
        H = a.shape[-1]
        a2 = _rows(_chk(a, "a"))
        assert a2.dtype == out_dtype, "embed_combine: `a` must have the compute dtype"
        rows, dev = a2.shape[0], a2.device
        f2 = _rows(f).float() if f is not None else None
        e2 = _rows(extra) if extra is not None else None
        y = _new((rows, H), out_dtype, dev)
        linb = _new((rows, H), out_dtype, dev) if f is not None else None
        xs = _new((rows, H), out_dtype, dev) if go is not None else None
        st = lambda on: (_new((rows,), torch.float32, dev), _new((rows,), torch.float32, dev)) if on else (None, None)
        ma, ra = st(ga is not None)
        mb, rb = st(f is not None)
        mo, ro = st(go is not None)
        dropping = p_drop > 0.0
        if not _ghost():
            row32 = row.reshape(-1) if row is not None else None
            if _INDEX_ERRORS[0] is None and (table is not None or table2 is not None) and not torch.cuda.is_current_stream_capturing():
                watch_index_errors(dev)
            _lib.call("vlni_embed_combine_fwd", _DT[out_dtype], a2.data_ptr(), a2.stride(0), _p(ga), _p(ba), _p(f2), f2.stride(0) if f2 is not None else 0,
                      f2.shape[1] if f2 is not None else 0, _p(Wb), _p(bb), _p(gb), _p(beb), _p(e2), e2.stride(0) if e2 is not None else 0, _p(row32),
                      _p(table), _p(idx), table.shape[0] if table is not None else 0, _p(table2), _p(idx2), table2.shape[0] if table2 is not None else 0,
                      _p(go), _p(bo), eps, _p(linb), _p(xs), y.data_ptr(), H, _p(ma), _p(ra), _p(mb), _p(rb),
                      _p(mo), _p(ro), p_drop if dropping else 0.0, _shift(seed, rows * H) if dropping else 0, rows, H, _st())
        ctx.save_for_backward(a2, f2, linb, xs, ma, ra, mb, rb, mo, ro, idx, idx2)
        ctx.P = (ga, ba, Wb, bb, gb, beb, row, table, table2, go, bo)
        ctx.cfg = (p_drop if dropping else 0.0, seed, a.shape, extra.shape if extra is not None else None, extra.dtype if extra is not None else None)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, dy):
        a2, f2, linb, xs, ma, ra, mb, rb, mo, ro, idx, idx2 = ctx.saved_tensors
        ga, ba, Wb, bb, gb, beb, row, table, table2, go, bo = ctx.P
        p_drop, seed, ashape, eshape, edtype = ctx.cfg
        ng = ctx.needs_input_grad
        H = a2.shape[1]
        d = _rows(dy)
        if p_drop > 0.0:
            d = dropout_apply(d, p_drop, seed)
        g = [None] * 11                                # gradients of ctx.P in order
        if go is not None:
            d, g[9], g[10] = _ln_bwd_to(d, xs, go, bo, mo, ro, ng[18] or ng[19])
        dsum = d
        da = dsum
        if ga is not None:
            da, g[0], g[1] = _ln_bwd_to(dsum, a2, ga, ba, ma, ra, ng[9] or ng[10])
        if f2 is not None:
            dl, g[4], g[5] = _ln_bwd_to(dsum, linb, gb, beb, mb, rb, ng[13] or ng[14])
            if ng[11] or ng[12]:
                N, K = Wb.shape
                if _direct(Wb) and (bb is None or _direct(bb)):
                    _lib.call("vlni_smallk_linear_bwd", _dt(dl), dl.data_ptr(), dl.stride(0), f2.data_ptr(), f2.stride(0), Wb.grad.data_ptr(),
                              _p(bb.grad if bb is not None else None), f2.shape[0], N, K, _st())
                else:
                    g[2] = torch.zeros_like(Wb)
                    g[3] = torch.zeros((N,), dtype=torch.float32, device=Wb.device) if bb is not None else None
                    _lib.call("vlni_smallk_linear_bwd", _dt(dl), dl.data_ptr(), dl.stride(0), f2.data_ptr(), f2.stride(0), g[2].data_ptr(), _p(g[3]),
                              f2.shape[0], N, K, _st())
        if row is not None and ng[15]:
            g[6] = colsum(dsum).view(row.shape).to(row.dtype)
        for k, (tab, ix) in ((7, (table, idx)), (8, (table2, idx2))):
            if tab is not None and ng[9 + k]:
                direct = _direct(tab) and tab.grad.is_contiguous()
                tg = tab.grad if direct else torch.zeros(tab.shape, dtype=torch.float32, device=dsum.device)
                if SMALL_TABLE_SCATTER and tab.shape[0] <= 8:
                    _lib.call("vlni_scatter_add_rows_small", _dt(dsum), dsum.data_ptr(), dsum.stride(0), ix.data_ptr(), tg.data_ptr(), dsum.shape[0], H,
                              tab.shape[0], _st())
                else:
                    _lib.call("vlni_scatter_add_rows", _dt(dsum), dsum.data_ptr(), dsum.stride(0), ix.data_ptr(), tg.data_ptr(), dsum.shape[0], H,
                              tab.shape[0], _st())
                g[k] = None if direct else tg
        d_extra = None
        if eshape is not None and ng[2]:
            d_extra = (dsum if dsum.dtype == edtype else cast(dsum, edtype)).view(eshape)
        return (da.view(ashape) if ng[0] else None, None, d_extra, None, None, None, None, None, None) + tuple(g)


def embed_combine(a, out_dtype, ln_a=None, small=None, row=None, table=None, table2=None, extra=None, ln_o=None, eps=1e-12, p_drop=0.0, training=False):
    """a: [.., H] activation in the compute dtype; ln_a = (gamma, beta); small = (features [.., K], W [H, K], bias, gamma, beta); row: a float32 [H] row;
    table / table2 = (float32 table, int64 row index per output row); extra: one more dense activation; ln_o = (gamma, beta) of the outer LayerNorm;
    dropout p_drop in training (counter-based mask: inside an episode tape the tape's seed, outside a fresh one)."""
    p = float(p_drop) if (training and p_drop > 0.0) else 0.0
    seed = 0
    if p > 0.0:
        seed = _TAPE.seed(1) if _TAPE is not None else next_seeds(1)
    if small is not None and small[0].shape[-1] > 4 and extra is None:
        # 7- / 14-d position features (DUET): a lane of the fused kernel would read 12 x K weights per row (its columns change with the lane, not
        # with the row) - 43 KB of W per row and wave at K = 14; the stand-alone small-K kernel keeps a column's weights in registers over the
        # rows. That branch stays two launches and joins as the dense extra source (measured: DUET 14.05 -> 14.38 ms with it fused).
        extra = layer_norm(smallk_linear(small[0], small[1], small[2], out_dtype), small[3], small[4], eps).reshape(a.shape)
        small = None
    f, Wb, bb, gb, beb = small if small is not None else (None,) * 5
    ga, ba = ln_a if ln_a is not None else (None, None)
    go, bo = ln_o if ln_o is not None else (None, None)
    tab, ix = table if table is not None else (None, None)
    tab2, ix2 = table2 if table2 is not None else (None, None)
    return _EmbedCombine.apply(a, f, extra, ix, ix2, eps, p, seed, out_dtype, ga, ba, Wb, bb, gb, beb, row, tab, tab2, go, bo)


class _SeqMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lens):
        B, S, H = x.shape
        x = _chk(x, "x").contiguous()
        if lens is not None:
            assert lens.shape == (B,) and lens.dtype == torch.int64 and lens.is_contiguous()
        out = _new((B, H), x.dtype, x.device)
        if not _ghost():
            _lib.call("vlni_seqmean_fwd", _dt(x), x.data_ptr(), out.data_ptr(), _p(lens), B, S, H, _st())
        ctx.dims, ctx.lens = (B, S, H), lens
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, H = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty((B, S, H), dtype=dout.dtype, device=dout.device)
        _lib.call("vlni_seqmean_bwd", _dt(dout), dout.data_ptr(), dx.data_ptr(), _p(ctx.lens), B, S, H, _st())
        return dx, None


class _RowDot(torch.autograd.Function):
    """logits[r] = mask[r] ? -inf : <h[r], w> + bias   (float32 logits)."""

    @staticmethod
    def forward(ctx, h, w, bias, mask):
        shp = h.shape
        h2 = _rows(_chk(h, "h"))
        rows, H = h2.shape
        m8 = _u8(mask).reshape(-1) if mask is not None else None
        out = _new((rows,), torch.float32, h.device)
        if not _ghost():
            _lib.call("vlni_rowdot_fwd", _dt(h2), h2.data_ptr(), h2.stride(0), w.data_ptr(), _p(bias), _p(m8), out.data_ptr(),
                      rows, H, _st())
        ctx.save_for_backward(h2, w, m8)
        ctx.shp, ctx.has_b, ctx.bias = shp, bias is not None, bias
        return out.view(shp[:-1])

    @staticmethod
    def backward(ctx, dl):
        h2, w, m8 = ctx.saved_tensors
        rows, H = h2.shape
        dl = dl.reshape(-1).float().contiguous()
        dh = torch.empty_like(h2)
        bias = ctx.bias
        if _direct(w) and (bias is None or _direct(bias)):        # the kernel adds (atomics) straight into the gradient arena
            _lib.call("vlni_rowdot_bwd", _dt(h2), dl.data_ptr(), h2.data_ptr(), h2.stride(0), w.data_ptr(), _p(m8),
                      dh.data_ptr(), dh.stride(0), w.grad.data_ptr(), _p(bias.grad if bias is not None else None), rows, H, _st())
            return dh.view(ctx.shp), None, None, None
        dw = torch.zeros((H,), dtype=torch.float32, device=h2.device)
        dbias = torch.zeros((1,), dtype=torch.float32, device=h2.device) if ctx.has_b else None
        _lib.call("vlni_rowdot_bwd", _dt(h2), dl.data_ptr(), h2.data_ptr(), h2.stride(0), w.data_ptr(), _p(m8),
                  dh.data_ptr(), dh.stride(0), dw.data_ptr(), _p(dbias), rows, H, _st())
        return dh.view(ctx.shp), dw.view(w.shape), dbias, None


class _LnRowDot(torch.autograd.Function):
    """logits[r] = mask[r] ? -inf : <dropout(LayerNorm(x[r])), w> + bias in ONE launch (vlni_ln_rowdot_fwd): the tail of NextActionPrediction
    (vilmodel_cmt.py:953-963,1200) and of DUET's ClsPrediction (vilmodel.py:1009-1020). Backward: the row dot's, the mask, LayerNorm's."""

    @staticmethod
    def forward(ctx, x, g, b, eps, p_drop, seed, w, bias, mask):
        shp = x.shape
        x2 = _rows(_chk(x, "x"))
        rows, H = x2.shape
        m8 = _u8(mask).reshape(-1) if mask is not None else None
        hd = _new((rows, H), x2.dtype, x.device)
        mean = _new((rows,), torch.float32, x.device)
        rstd = _new((rows,), torch.float32, x.device)
        out = _new((rows,), torch.float32, x.device)
        if not _ghost():
            _lib.call("vlni_ln_rowdot_fwd", _dt(x2), x2.data_ptr(), x2.stride(0), g.data_ptr(), b.data_ptr(), eps, hd.data_ptr(), mean.data_ptr(),
                      rstd.data_ptr(), w.data_ptr(), _p(bias), _p(m8), out.data_ptr(), p_drop, _shift(seed, rows * H) if p_drop > 0.0 else 0, rows, H, _st())
        ctx.save_for_backward(x2, hd, mean, rstd, w, m8)
        ctx.P, ctx.cfg = (g, b, bias), (shp, p_drop, seed)
        return out.view(shp[:-1])

    @staticmethod
    def backward(ctx, dl):
        x2, hd, mean, rstd, w, m8 = ctx.saved_tensors
        g, b, bias = ctx.P
        shp, p_drop, seed = ctx.cfg
        ng = ctx.needs_input_grad
        rows, H = x2.shape
        dl = dl.reshape(-1).float().contiguous()
        dh = torch.empty_like(hd)
        dw = dbias = None
        if _direct(w) and (bias is None or _direct(bias)):
            _lib.call("vlni_rowdot_bwd", _dt(hd), dl.data_ptr(), hd.data_ptr(), hd.stride(0), w.data_ptr(), _p(m8), dh.data_ptr(), dh.stride(0),
                      w.grad.data_ptr(), _p(bias.grad if bias is not None else None), rows, H, _st())
        else:
            dw = torch.zeros((H,), dtype=torch.float32, device=hd.device)
            dbias = torch.zeros((1,), dtype=torch.float32, device=hd.device) if bias is not None else None
            _lib.call("vlni_rowdot_bwd", _dt(hd), dl.data_ptr(), hd.data_ptr(), hd.stride(0), w.data_ptr(), _p(m8), dh.data_ptr(), dh.stride(0),
                      dw.data_ptr(), _p(dbias), rows, H, _st())
            dw = dw.view(w.shape)
        if p_drop > 0.0:
            dh = dropout_apply(dh, p_drop, seed)
        dx, dg, db = _ln_bwd_to(dh, x2, g, b, mean, rstd, ng[1] or ng[2])
        return dx.view(shp) if ng[0] else None, dg, db, None, None, None, dw, dbias, None


def ln_rowdot(x, g, b, w, bias, mask, eps=1e-12, p_drop=0.0, training=False):
    p = float(p_drop) if (training and p_drop > 0.0) else 0.0
    seed = (_TAPE.seed(1) if _TAPE is not None else next_seeds(1)) if p > 0.0 else 0
    return _LnRowDot.apply(x, g, b, eps, p, seed, w, bias, mask)


class _DuetFuse(torch.autograd.Function):
    """fused = global + local logits scattered onto map nodes (plan = (src [B,G] int32, bw [B,V] uint8))."""

    @staticmethod
    def forward(ctx, gl, ll, src, bw):
        gl, ll = _chk(gl, "global_logits").float().contiguous(), ll.float().contiguous()
        (B, G), V = gl.shape, ll.shape[1]
        assert src.shape == (B, G) and bw.shape == (B, V) and src.dtype == torch.int32 and bw.dtype == torch.uint8
        out = _new(gl.shape, gl.dtype, gl.device)
        if not _ghost():
            _lib.call("vlni_duet_fuse_fwd", gl.data_ptr(), ll.data_ptr(), src.data_ptr(), bw.data_ptr(), out.data_ptr(), B, G, V, _st())
        ctx.save_for_backward(src, bw)
        return out

    @staticmethod
    def backward(ctx, dout):
        src, bw = ctx.saved_tensors
        dout = dout.float().contiguous()
        (B, G), V = dout.shape, bw.shape[1]
        dll = torch.empty((B, V), dtype=torch.float32, device=dout.device)
        _lib.call("vlni_duet_fuse_bwd", dout.data_ptr(), src.data_ptr(), bw.data_ptr(), dll.data_ptr(), B, G, V, _st())
        return dout, dll, None, None


def duet_fuse(gl, ll, src, bw):
    return _DuetFuse.apply(gl, ll, src, bw)


def _u8(m):
    """bool / uint8 mask as contiguous bytes (a bool tensor is reinterpreted, not copied)."""
    m = m.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


class _DuetHeads(torch.autograd.Function):
    """(global_logits, local_logits, fused_logits) of a DUET navigation call from the two heads' raw outputs, the pre-sigmoid fuse
    weight, the masks and the fusion plan: ONE launch per direction (vlni_duet_heads_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, graw, lraw, f, visited, gmask, nav, src, bw):
        graw, lraw = _chk(graw, "global head").float().contiguous(), lraw.float().contiguous()
        (B, G), V = graw.shape, lraw.shape[1]
        assert src.shape == (B, G) and bw.shape == (B, V) and src.dtype == torch.int32 and bw.dtype == torch.uint8
        assert visited.shape == (B, G) and gmask.shape == (B, G) and nav.shape == (B, V)
        if f is not None:
            f = f.float().reshape(B).contiguous()
        vis8, gm8, nav8 = _u8(visited), _u8(gmask), _u8(nav)
        gl, ll, fu = _new((B, G), torch.float32, graw.device), _new((B, V), torch.float32, graw.device), _new((B, G), torch.float32, graw.device)
        if not _ghost():
            _lib.call("vlni_duet_heads_fwd", graw.data_ptr(), lraw.data_ptr(), _p(f), vis8.data_ptr(), gm8.data_ptr(), nav8.data_ptr(),
                      src.data_ptr(), bw.data_ptr(), gl.data_ptr(), ll.data_ptr(), fu.data_ptr(), B, G, V, _st())
        ctx.save_for_backward(graw, lraw, f, vis8, gm8, nav8, src, bw)
        ctx.set_materialize_grads(False)                # an unused output's gradient arrives as None, not as a zero tensor
        return gl, ll, fu

    @staticmethod
    def backward(ctx, d_gl, d_ll, d_fu):
        graw, lraw, f, vis8, gm8, nav8, src, bw = ctx.saved_tensors
        (B, G), V = graw.shape, lraw.shape[1]
        c = lambda d: None if d is None else d.float().contiguous()
        d_gl, d_ll, d_fu = c(d_gl), c(d_ll), c(d_fu)
        dgraw, dlraw = torch.empty_like(graw), torch.empty_like(lraw)
        df = torch.empty((B,), dtype=torch.float32, device=graw.device) if f is not None else None
        _lib.call("vlni_duet_heads_bwd", _p(d_gl), _p(d_ll), _p(d_fu), graw.data_ptr(), lraw.data_ptr(), _p(f), vis8.data_ptr(),
                  gm8.data_ptr(), nav8.data_ptr(), src.data_ptr(), bw.data_ptr(), dgraw.data_ptr(), dlraw.data_ptr(), _p(df), B, G, V, _st())
        return dgraw, dlraw, df, None, None, None, None, None


def duet_heads(graw, lraw, f, visited, gmask, nav, src, bw):
    """f: [B] / [B,1] pre-sigmoid fuse weight or None (fixed 0.5, glocal_fuse off)."""
    return _DuetHeads.apply(graw, lraw, f, visited, gmask, nav, src, bw)


class _CrossEntropySum(torch.autograd.Function):
    """CrossEntropyLoss(ignore_index=-100, reduction='sum') on float32 logits (may hold -inf)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lg = _chk(logits, "logits").float().contiguous()
        rows, V = lg.shape
        loss = torch.zeros((1,), dtype=torch.float32, device=lg.device)
        dlg = torch.empty_like(lg)
        _lib.call("vlni_cross_entropy", lg.data_ptr(), lg.stride(0), target.data_ptr(), ignore_index, loss.data_ptr(),
                  dlg.data_ptr(), dlg.stride(0), rows, V, _st())
        ctx.save_for_backward(dlg)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlg,) = ctx.saved_tensors
        return dlg * g, None, None


class _SegmentMean(torch.autograd.Function):
    """out[s] = mean of x rows listed in CSR (seg_off, rowidx): noun-phrase token mean (vilmodel_cmt.py:766-779)."""

    @staticmethod
    def forward(ctx, x2, seg_off, rowidx):
        _chk(x2, "x")
        nseg, H = seg_off.numel() - 1, x2.shape[1]
        out = torch.empty((nseg, H), dtype=x2.dtype, device=x2.device)
        _lib.call("vlni_segment_mean_fwd", _dt(x2), x2.data_ptr(), x2.stride(0), seg_off.data_ptr(), rowidx.data_ptr(),
                  out.data_ptr(), nseg, H, _st())
        ctx.save_for_backward(seg_off, rowidx)
        ctx.xshape, ctx.xdtype = x2.shape, x2.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        seg_off, rowidx = ctx.saved_tensors
        dout = dout.contiguous()
        nseg, H = dout.shape
        dx32 = torch.zeros(ctx.xshape, dtype=torch.float32, device=dout.device)
        _lib.call("vlni_segment_mean_bwd", _dt(dout), dout.data_ptr(), seg_off.data_ptr(), rowidx.data_ptr(),
                  dx32.data_ptr(), nseg, H, _st())
        return cast(dx32, ctx.xdtype), None, None


class _Cosine(torch.autograd.Function):
    """cos[r] = cosine_similarity(x[r], y[r], eps=1e-8) -> float32 [rows]."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _chk(x, "x").contiguous(), y.contiguous()
        rows, H = x.shape
        cosv = torch.empty((rows,), dtype=torch.float32, device=x.device)
        nx, ny = torch.empty_like(cosv), torch.empty_like(cosv)
        _lib.call("vlni_cosine_fwd", _dt(x), x.data_ptr(), y.data_ptr(), 1e-8, cosv.data_ptr(), nx.data_ptr(), ny.data_ptr(),
                  rows, H, _st())
        ctx.save_for_backward(x, y, cosv, nx, ny)
        return cosv

    @staticmethod
    def backward(ctx, g):
        x, y, cosv, nx, ny = ctx.saved_tensors
        rows, H = x.shape
        g = g.float().contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        _lib.call("vlni_cosine_bwd", _dt(x), x.data_ptr(), y.data_ptr(), g.data_ptr(), cosv.data_ptr(), nx.data_ptr(),
                  ny.data_ptr(), _p(dx), _p(dy), rows, H, _st())
        return dx, dy


class _PairDot(torch.autograd.Function):
    """out[i, j] = <a_i, b_j> for two small float32 row sets (vlni_pairdot_*): the similarity matrix of the alignment head's
    in-batch-negatives losses (vilmodel_cmt.py:793-856)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "a").float().contiguous(), _chk(b, "b").float().contiguous()
        (na, H), nb = a.shape, b.shape[0]
        out = torch.empty((na, nb), dtype=torch.float32, device=a.device)
        _lib.call("vlni_pairdot_fwd", a.data_ptr(), b.data_ptr(), out.data_ptr(), na, nb, H, _st())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.float().contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.call("vlni_pairdot_bwd", g.data_ptr(), a.data_ptr(), b.data_ptr(), _p(da), _p(db), a.shape[0], b.shape[0], a.shape[1], _st())
        return da, db


def pairdot(a, b):
    return _PairDot.apply(a, b)


class _GateRows(torch.autograd.Function):
    """f = visn[:, r0:r0+n] * lang[:, :1]  (NavCMT's action-head input for act_pred_token == 'ob_txt', vilmodel_cmt.py:1192):
    one kernel forward and one backward (vlni_gate_rows_*) instead of two slices, a broadcast multiply and their backward (zero-filled
    full-size gradients, casts, a reduction: ~14 launches per navigation step)."""

    @staticmethod
    def forward(ctx, visn, lang, r0, n):
        visn, lang = _chk(visn, "visn").contiguous(), _chk(lang, "lang").contiguous()
        (B, Sv, H), Sl = visn.shape, lang.shape[1]
        f = _new((B, n, H), visn.dtype, visn.device)
        if not _ghost():
            _lib.call("vlni_gate_rows_fwd", _dt(visn), visn.data_ptr(), lang.data_ptr(), f.data_ptr(), B, Sv, Sl, r0, n, H, _st())
        ctx.save_for_backward(visn, lang)
        ctx.meta = (r0, n)
        return f

    @staticmethod
    def backward(ctx, df):
        visn, lang = ctx.saved_tensors
        r0, n = ctx.meta
        (B, Sv, H), Sl = visn.shape, lang.shape[1]
        dvisn = torch.empty_like(visn) if ctx.needs_input_grad[0] else None       # the kernel writes every row of both
        dlang = torch.empty_like(lang) if ctx.needs_input_grad[1] else None
        if dvisn is not None or dlang is not None:
            _lib.call("vlni_gate_rows_bwd", _dt(visn), _chk(df, "df").contiguous().data_ptr(), visn.data_ptr(), lang.data_ptr(), _p(dvisn), _p(dlang),
                      B, Sv, Sl, r0, n, H, _st())
        return dvisn, dlang, None, None


def gate_rows(visn, lang, r0, n):
    return _GateRows.apply(visn, lang, r0, n)


# ---- functional front-ends ------------------------------------------------------------
NO_DROP = (0.0, 0.0, 0)


class _RepeatUnread(torch.autograd.Function):
    """x [n, ...] -> [T * n, ...] (T copies) for a tensor that the GHOST pass of an episode tape hands to operators which launch nothing and whose
    backward never reads it (a residual input): inside a ghost pass the copies are not made (uninitialised memory of the right shape - 63 MB
    of clone per episode for HAMT's language stream), anywhere else this is expand + reshape. Backward: the sum over the T copies."""

    @staticmethod
    def forward(ctx, x, T):
        ctx.T = T
        shape = (T * x.shape[0],) + tuple(x.shape[1:])
        if _ghost():
            return torch.empty(shape, dtype=x.dtype, device=x.device)
        return x.unsqueeze(0).expand((T,) + tuple(x.shape)).reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        return dy.reshape((ctx.T, dy.shape[0] // ctx.T) + tuple(dy.shape[1:])).sum(0), None


def repeat_unread(x, T):
    return _RepeatUnread.apply(x, T)


def drop_cfg(p_attn, p_hidden, training):
    """(p_attn, p_hidden, seed) for one fused block call; zeros outside training."""
    if not training or (p_attn <= 0.0 and p_hidden <= 0.0):
        return NO_DROP
    return (float(p_attn), float(p_hidden), _TAPE.seed(4) if _TAPE is not None else next_seeds(4))


class _TapeDropout(torch.autograd.Function):
    """Counter-based dropout (vlni_dropout) as an autograd node: what `dropout` uses inside an episode tape, where torch's own generator
    would draw one mask in the step's forward and another one in the ghost pass."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _chk(x, "x").contiguous()
        y = _new(x.shape, x.dtype, x.device)
        if not _ghost():
            _lib.call("vlni_dropout", _dt(x), x.data_ptr(), y.data_ptr(), x.numel(), p, _shift(seed, x.numel()), _st())
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        return dropout_apply(dy, ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    """F.dropout for the small tensors the models drop outside the fused kernels (embedding outputs, action heads); inside an episode
    tape the mask comes from the library's counter-based hash so that the ghost pass and the batched backward see the step's mask."""
    if not training or p <= 0.0:
        return x
    if _TAPE is None:
        return torch.nn.functional.dropout(x, p, True)
    return _TapeDropout.apply(x, float(p), _TAPE.seed(1))


def self_att_block(x, kmask, p, eps=1e-12, bias=None, drop=NO_DROP):
    return _SelfAttBlock.apply(x, kmask, bias, eps, drop, *p)


def ffn_block(x, p, eps=1e-12, drop=NO_DROP):
    return _FfnBlock.apply(x, eps, drop, *p)


def dual_self_att_block(x0, x1, km0, km1, p0, p1, eps=1e-12, drop0=NO_DROP, drop1=NO_DROP, bias0=None):
    """bias0: optional additive [B, S0, S0] attention bias of stream 0 (DUET's graph_sprels), differentiable."""
    return _DualSelfAttBlock.apply(x0, x1, km0, km1, bias0, eps, drop0, drop1, *p0, *p1)


def dual_xatt_q_block(x0, x1, kv0, kv1, mask_c, p0, p1, eps=1e-12, drop0=NO_DROP, drop1=NO_DROP):
    sel = lambda p: (p[0], p[1], p[6], p[7], p[8], p[9])
    return _DualXAttQBlock.apply(x0, x1, kv0, kv1, mask_c, eps, drop0, drop1, *sel(p0), *sel(p1))


def dual_ffn_block(x0, x1, p0, p1, eps=1e-12, drop0=NO_DROP, drop1=NO_DROP):
    return _DualFfnBlock.apply(x0, x1, eps, drop0, drop1, *p0, *p1)


def xatt_pair_block(lang, visn, mask_l, mask_v, p, eps=1e-12, drop=NO_DROP):
    return _XAttPairBlock.apply(lang, visn, mask_l, mask_v, eps, drop, *p)


def qkv_proj(x, p):
    """Packed Q / K / V projection of one stream; p = (wq,bq,wk,bk,wv,bv,wo,bo,g,b) of the attention holder. [B*S, 3H]."""
    return _QkvProj.apply(x, *p[:6])


def xatt_pair_given_q_block(lang, lang_qkv, visn, mask_l, mask_v, p, eps=1e-12, drop=NO_DROP):
    return _XAttPairGivenQBlock.apply(lang, lang_qkv, visn, mask_l, mask_v, eps, drop, *p)


def xatt_block(x, c_in, mask_c, p, eps=1e-12, drop=NO_DROP):
    return _XAttBlock.apply(x, c_in, mask_c, eps, drop, *p)


def kv_proj(c_in, p):
    """Packed K/V projection of a cross-attention context; p = (wq,bq,wk,bk,wv,bv,wo,bo,g,b) of the attention holder."""
    return _KvProj.apply(c_in, p[2], p[3], p[4], p[5])


def xatt_q_block(x, kv, mask_c, p, eps=1e-12, drop=NO_DROP):
    return _XAttQBlock.apply(x, kv, mask_c, eps, drop, p[0], p[1], p[6], p[7], p[8], p[9])


def prenorm_att_block(x, kmask, p, eps=1e-5, drop=NO_DROP):
    return _PreNormAttBlock.apply(x, kmask, eps, drop, *p)


def prenorm_ffn_block(x, p, eps=1e-5, drop=NO_DROP):
    return _PreNormFfnBlock.apply(x, eps, drop, *p)


def linear(x, w, b=None, act=0, out_dtype=None):
    return _Linear.apply(x, w, b, act, out_dtype or x.dtype)


def layer_norm(x, g, b, eps=1e-12):
    return _LayerNorm.apply(x, g, b, eps)


def smallk_linear(x, w, b, out_dtype):
    return _SmallKLinear.apply(x, w, b, out_dtype)


def sum_layer_norm(srcs, g, b, rows, out_dtype, eps=1e-12):
    """srcs: list of (tensor, kind, idx) with kind in {'dense','bcast','gather'}; idx int64 [rows] for 'gather'."""
    spec = tuple((k, False) for _, k, _ in srcs)
    idxs = tuple(i for _, _, i in srcs)
    return _SumLayerNorm.apply(spec, idxs, eps, out_dtype, rows, g, b, *[t for t, _, _ in srcs])


def seq_mean(x, lens=None):
    """[B,S,H] -> [B,H] mean over the sequence; lens (int64 [B]): over each sample's first lens[b] rows only."""
    return _SeqMean.apply(x, lens)


def row_dot(h, w, bias, mask):
    return _RowDot.apply(h, w, bias, mask)


def cross_entropy_sum(logits, target, ignore_index=-100):
    return _CrossEntropySum.apply(logits, target, ignore_index)


def segment_mean(x2, seg_off, rowidx):
    return _SegmentMean.apply(x2, seg_off, rowidx)


def cosine(x, y):
    return _Cosine.apply(x, y)
