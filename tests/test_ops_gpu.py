"""GPU: every HIP operator (through the C-ABI) against a plain PyTorch fp32 reference of the same op.
fp32 path tolerance 1e-4 (BASELINE north_star); bf16 path is checked against the fp32 math on
bf16-rounded inputs with a bf16-sized tolerance (never used for the parity gate)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from vln_imagine_amd import ops as o
    return o


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _err(a, b):
    return (a.double() - b.double()).abs().max().item()


# max |error| allowed, in units of max(1, max |reference|): float32 = the parity gate's 1e-4; the 16-bit paths are compared with float64 math on the
# SAME 16-bit inputs, so what is left is output rounding (2^-9 bfloat16, 2^-12 float16 relative) plus the 16-bit intermediates of the fused chains
TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}


def _chk(a, ref, t, what=""):
    e, lim = _err(a, ref), t * max(1.0, ref.abs().max().item())
    assert e < lim, (what, e, lim)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (148, 768, 768), (300, 2304, 768), (257, 768, 3072), (64, 512, 512),
                                   (5, 1536, 768)])
def test_gemm_nt_plain(ops, dtype, M, N, K):
    a, b = _rand((M, K), dtype, 1, 0.5), _rand((N, K), dtype, 2, 0.05)
    ref = a.double() @ b.double().t()
    out = ops.gemm_nt(a, b)
    assert out.dtype == dtype
    e = _err(out, ref)
    assert e < TOL[dtype] * max(1.0, ref.abs().max().item()), (M, N, K, e)


def test_gemm_exact_integers(ops):
    """asymmetric small-integer operands: any row/col or k-permutation slip shows as a whole-number error."""
    M, N, K = 192, 256, 128
    a = torch.randint(-3, 4, (M, K), generator=torch.Generator().manual_seed(3)).float()
    b = torch.randint(-3, 4, (N, K), generator=torch.Generator().manual_seed(4)).float()
    ref = a @ b.t()
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        out = ops.gemm_nt(a.to(dtype).cuda(), b.to(dtype).cuda()).float().cpu()
        assert torch.equal(out, ref), dtype


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 200, 768, 768
    a, b = _rand((M, K), dtype, 1, 0.5), _rand((N, K), dtype, 2, 0.05)
    bias = _rand((N,), torch.float32, 3, 0.1)
    res = _rand((M, N), dtype, 4, 0.5)
    z = torch.empty((M, N), dtype=dtype, device="cuda")
    out = ops.gemm_nt(a, b, bias=bias, act=1, residual=res, preact=z)
    zr = a.double() @ b.double().t() + bias.double()
    ref = torch.nn.functional.gelu(zr) + res.double()
    assert _err(z, zr) < TOL[dtype] * 3
    assert _err(out, ref) < TOL[dtype] * 3
    # relu + dact (gelu') epilogue
    out2 = ops.gemm_nt(a, b, dact_src=z, dact=1)
    zz = z.double().requires_grad_(True)
    gp = torch.autograd.grad(torch.nn.functional.gelu(zz).sum(), zz)[0]
    ref2 = (a.double() @ b.double().t()) * gp
    assert _err(out2, ref2) < TOL[dtype] * 3
    out3 = ops.gemm_nt(a, b, bias=bias, act=2)
    assert _err(out3, torch.relu(zr)) < TOL[dtype] * 3
    # act 3: GELU with GELU'(pre) stored in place of pre; dact 3: multiply by a stored derivative (the FFN pair of the 16-bit paths)
    g = torch.empty((M, N), dtype=dtype, device="cuda")
    out4 = ops.gemm_nt(a, b, bias=bias, act=3, residual=res, preact=g)
    zz = zr.clone().requires_grad_(True)
    gref = torch.autograd.grad(torch.nn.functional.gelu(zz).sum(), zz)[0]
    _chk(out4, ref, TOL[dtype], "act 3 output")
    _chk(g, gref, TOL[dtype], "act 3 stored derivative")
    out5 = ops.gemm_nt(a, b, dact_src=g, dact=3, residual=res)
    _chk(out5, (a.double() @ b.double().t()) * g.double() + res.double(), TOL[dtype], "dact 3")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(148, 768, 768), (5504, 768, 768), (333, 3072, 768), (70, 768, 3072)])
def test_wgrad(ops, dtype, M, N, K):
    dy, x = _rand((M, N), dtype, 5, 0.1), _rand((M, K), dtype, 6, 0.5)
    ref = dy.double().t() @ x.double()
    out = ops.wgrad(dy, x)
    assert out.dtype == torch.float32
    tol = 1e-4 if dtype == torch.float32 else 2e-3
    assert _err(out, ref) < tol * max(1.0, ref.abs().max().item()), _err(out, ref)
    out2 = ops.wgrad(dy, x, out)                       # accumulate
    assert _err(out2, 2 * ref) < 2 * tol * max(1.0, ref.abs().max().item())
    cs = ops.colsum(dy)
    assert _err(cs, dy.double().sum(0)) < tol * 10


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,H,eps", [(148, 768, 1e-12), (1, 768, 1e-12), (4099, 768, 1e-5), (37, 512, 1e-12)])
def test_layernorm(ops, dtype, rows, H, eps):
    x = _rand((rows, H), dtype, 7)
    g, b = _rand((H,), torch.float32, 8, 0.1) + 1.0, _rand((H,), torch.float32, 9, 0.1)
    dy = _rand((rows, H), dtype, 10)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), gr, br, eps)
    yr.backward(dy.double())
    y, mean, rstd = ops.ln_fwd(x, g, b, eps)
    dx, dg, db = ops.ln_bwd(dy, x, g, mean, rstd)
    t = TOL[dtype]
    _chk(y, yr, t, "y")
    _chk(dx, xr.grad, t, "dx")
    assert _err(dg, gr.grad) < t * max(1.0, gr.grad.abs().max().item())
    assert _err(db, br.grad) < t * max(1.0, br.grad.abs().max().item())


def _attn_ref(q, k, v, kmask, bias, B, Sq, Sk):
    nh, dh = 12, 64
    qq = q.double().view(B, Sq, nh, dh).transpose(1, 2)
    kk = k.double().view(B, Sk, nh, dh).transpose(1, 2)
    vv = v.double().view(B, Sk, nh, dh).transpose(1, 2)
    s = qq @ kk.transpose(-1, -2) / 8.0
    if kmask is not None:
        s = s + kmask.double()[:, None, None, :]
    if bias is not None:
        s = s + bias.double()[:, None]
    return (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B * Sq, nh * dh)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Sq,Sk,use_bias", [(4, 80, 80, False), (3, 86, 38, False), (3, 38, 86, False), (2, 1, 5, False),
                                              (2, 100, 128, False), (2, 33, 65, True), (5, 130, 97, False)])
def test_attention_fwd_bwd(ops, dtype, B, Sq, Sk, use_bias):
    H = 768
    qkv_q = _rand((B * Sq, 3 * H), dtype, 11, 0.7)          # packed: q taken from cols 0:768
    qkv_k = _rand((B * Sk, 3 * H), dtype, 12, 0.7)          # k from 768:1536, v from 1536:2304
    lens = torch.randint(1, Sk + 1, (B,), generator=torch.Generator().manual_seed(13))
    lens[0] = Sk
    kmask = ((torch.arange(Sk)[None, :] >= lens[:, None]).float() * -10000.0).cuda()
    bias = _rand((B, Sq, Sk), torch.float32, 14, 0.5) if use_bias else None
    q, k, v = qkv_q[:, :H], qkv_k[:, H:2 * H], qkv_k[:, 2 * H:]
    out, lse = ops.attn_fwd(q, k, v, B, Sq, Sk, kmask, bias)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    br = bias.double().clone().requires_grad_(True) if use_bias else None
    ref = _attn_ref(qr, kr, vr, kmask, br, B, Sq, Sk)
    t = TOL[dtype]
    _chk(out, ref, t, "fwd")
    dout = _rand((B * Sq, H), dtype, 15)
    ref.backward(dout.double())
    dq_buf = torch.zeros_like(qkv_q)
    dk_buf = torch.zeros_like(qkv_k)
    dbias = torch.zeros_like(bias) if use_bias else None
    ops.attn_bwd(q, k, v, out, dout, lse, dq_buf[:, :H], dk_buf[:, H:2 * H], dk_buf[:, 2 * H:], B, Sq, Sk, kmask, bias, dbias)
    _chk(dq_buf[:, :H], qr.grad, t, "dq")
    _chk(dk_buf[:, H:2 * H], kr.grad, t, "dk")
    _chk(dk_buf[:, 2 * H:], vr.grad, t, "dv")
    assert float(dq_buf[:, H:].abs().max()) == 0.0 and float(dk_buf[:, :H].abs().max()) == 0.0   # nothing outside the slices
    if use_bias:
        _chk(dbias, br.grad, t * 5, "dbias")


@pytest.mark.parametrize("B,Sq,Sk", [(2, 45, 150), (2, 70, 200), (1, 130, 256), (2, 37, 129)])
def test_attention_long_context_bf16(ops, B, Sq, Sk):
    """129..256 keys (long DUET instructions + imaginations): bf16 kernels, 8-wave backward."""
    H, dtype = 768, torch.bfloat16
    qt, kt = _rand((B * Sq, H), dtype, 21, 0.7), _rand((B * Sk, 2 * H), dtype, 22, 0.7)
    lens = torch.tensor([Sk] + [Sk - 37] * (B - 1))
    kmask = ((torch.arange(Sk)[None, :] >= lens[:, None]).float() * -10000.0).cuda()
    k, v = kt[:, :H], kt[:, H:]
    out, lse = ops.attn_fwd(qt, k, v, B, Sq, Sk, kmask)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (qt, k, v))
    ref = _attn_ref(qr, kr, vr, kmask, None, B, Sq, Sk)
    assert _err(out, ref) < 3e-2, _err(out, ref)
    dout = _rand((B * Sq, H), dtype, 23)
    ref.backward(dout.double())
    dq, dkv = torch.zeros_like(qt), torch.zeros_like(kt)
    ops.attn_bwd(qt, k, v, out, dout, lse, dq, dkv[:, :H], dkv[:, H:], B, Sq, Sk, kmask)
    for a, r, n in ((dq, qr.grad, "dq"), (dkv[:, :H], kr.grad, "dk"), (dkv[:, H:], vr.grad, "dv")):
        assert _err(a, r) < 6e-2, (n, _err(a, r))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Sq,Sk,use_bias", [(2, 45, 206, False), (2, 206, 206, True), (1, 70, 300, False), (2, 5, 513, True)])
def test_attention_beyond_the_tile_kernels(ops, dtype, B, Sq, Sk, use_bias):
    """More keys than the tile kernels take (128 in float32 - the reference's DUET scripts run --max_instr_len 200 on the parity path -
    256 in the 16-bit types): the generic kernels behind them, against float64 torch incl. the bias gradient."""
    if dtype == torch.bfloat16 and Sk <= 256:
        pytest.skip("the tile kernels' range")
    H = 768
    qt, kt = _rand((B * Sq, H), dtype, 21, 0.7), _rand((B * Sk, 2 * H), dtype, 22, 0.7)
    lens = torch.tensor([Sk] + [Sk - 37] * (B - 1))
    kmask = ((torch.arange(Sk)[None, :] >= lens[:, None]).float() * -10000.0).cuda()
    bias = _rand((B, Sq, Sk), torch.float32, 24, 0.5) if use_bias else None
    k, v = kt[:, :H], kt[:, H:]
    out, lse = ops.attn_fwd(qt, k, v, B, Sq, Sk, kmask, bias)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (qt, k, v))
    br = bias.double().clone().requires_grad_(True) if use_bias else None
    ref = _attn_ref(qr, kr, vr, kmask, br, B, Sq, Sk)
    t = 2e-5 if dtype == torch.float32 else 3e-2
    assert _err(out, ref) < t, _err(out, ref)
    dout = _rand((B * Sq, H), dtype, 23)
    ref.backward(dout.double())
    dq, dkv = torch.zeros_like(qt), torch.zeros_like(kt)
    dbias = torch.zeros_like(bias) if use_bias else None
    ops.attn_bwd(qt, k, v, out, dout, lse, dq, dkv[:, :H], dkv[:, H:], B, Sq, Sk, kmask, bias, dbias)
    for a, r, n in ((dq, qr.grad, "dq"), (dkv[:, :H], kr.grad, "dk"), (dkv[:, H:], vr.grad, "dv")):
        assert _err(a, r) < 2 * t, (n, _err(a, r))
    if use_bias:
        assert _err(dbias, br.grad) < 5 * t, _err(dbias, br.grad)
    with pytest.raises(Exception, match="not covered"):
        ops.attn_fwd(qt, k[:1].expand(B * 2100, H).contiguous(), v[:1].expand(B * 2100, H).contiguous(), B, Sq, 2100, None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blocks_against_torch(ops, dtype):
    """The fused sublayer nodes (self-att, FFN, x-att pair, x-att) fwd+bwd vs the same math in torch fp64."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, Sl, Sv, H, FF = 3, 50, 21, 768, 3072
    mk = lambda *s, sc=0.04: (torch.randn(*s) * sc).cuda().requires_grad_(True)
    att = [mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1)]
    g = (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True)
    b = (0.1 * torch.randn(H)).cuda().requires_grad_(True)
    ffn = [mk(FF, H), mk(FF, sc=0.1), mk(H, FF, sc=0.02), mk(H, sc=0.1)]
    lang = torch.randn(B, Sl, H).cuda()
    visn = torch.randn(B, Sv, H).cuda()
    ml = (torch.rand(B, Sl) > 0.2).float()
    mv = (torch.rand(B, Sv) > 0.2).float()
    ml[:, 0] = 1; mv[:, 0] = 1
    aml, amv = ((1 - ml) * -10000.0).cuda(), ((1 - mv) * -10000.0).cuda()

    def ref_att(x, c, mask, W):
        wq, bq, wk, bk, wv, bv, wo, bo = [w.double() for w in W]
        Bq, Sq, _ = x.shape
        Sk = c.shape[1]
        q = F.linear(x, wq, bq).view(Bq, Sq, 12, 64).transpose(1, 2)
        k = F.linear(c, wk, bk).view(Bq, Sk, 12, 64).transpose(1, 2)
        v = F.linear(c, wv, bv).view(Bq, Sk, 12, 64).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / 8 + mask.double()[:, None, None, :]
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(Bq, Sq, H)
        return F.layer_norm(F.linear(a, wo, bo) + x, (H,), g.double(), b.double(), 1e-12)

    def ref_ffn(x):
        w1, b1, w2, b2 = [w.double() for w in ffn]
        return F.layer_norm(F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) + x, (H,), g.double(), b.double(), 1e-12)

    params = att + [g, b] + ffn
    P = tuple(att) + (g, b)
    PF = tuple(ffn) + (g, b)

    def run(fn_prod, fn_ref, inputs):
        xs_p = [t.to(dtype).detach().requires_grad_(True) for t in inputs]
        xs_r = [t.to(dtype).double().detach().requires_grad_(True) for t in inputs]
        for p in params:
            p.grad = None
        out_p = fn_prod(*xs_p)
        out_r = fn_ref(*xs_r)
        out_p = out_p if isinstance(out_p, tuple) else (out_p,)
        out_r = out_r if isinstance(out_r, tuple) else (out_r,)
        t = TOL[dtype]
        for a_, r_ in zip(out_p, out_r):
            _chk(a_, r_, t, "fwd")
        torch.manual_seed(1)
        ws = [torch.randn_like(r_) for r_ in out_r]
        sum((a_.double() * w).sum() for a_, w in zip(out_p, ws)).backward()
        gp = [p.grad.clone() for p in params if p.grad is not None]
        gx = [x.grad.clone() for x in xs_p]
        for p in params:
            p.grad = None
        sum((r_ * w).sum() for r_, w in zip(out_r, ws)).backward()
        gr = [p.grad.clone() for p in params if p.grad is not None]
        scale = 4.0 if dtype == torch.bfloat16 else 1.0      # a chain of 16-bit intermediates (measured: <= 2.6 x the single-op bound)
        for a_, r_ in zip(gx, [x.grad for x in xs_r]):
            assert _err(a_, r_) < t * scale * max(1.0, r_.abs().max().item()), ("dx", _err(a_, r_))
        assert len(gp) == len(gr)
        for a_, r_ in zip(gp, gr):
            if dtype != torch.float32 and r_.abs().max().item() < 1e-9:
                # the key bias: softmax is shift-invariant, the exact gradient is 0 and the 16-bit one is what is left when B x S rounded dK rows
                # are summed - rounding noise of that sum (measured 0.06), not a fraction of the (zero) reference
                assert _err(a_, r_) < 0.1, ("dparam with a zero reference", _err(a_, r_))
                continue
            assert _err(a_, r_) < t * scale * max(1.0, r_.abs().max().item()), ("dparam", _err(a_, r_), r_.abs().max().item())

    run(lambda x: ops.self_att_block(x, aml, P), lambda x: ref_att(x, x, aml, att), [lang])
    run(lambda x: ops.ffn_block(x, PF), ref_ffn, [lang])
    run(lambda l, v: ops.xatt_pair_block(l, v, aml, amv, P),
        lambda l, v: (ref_att(l, v, amv, att), ref_att(v, l, aml, att)), [lang, visn])
    run(lambda v, l: ops.xatt_block(v, l, aml, P), lambda v, l: ref_att(v, l, aml, att), [visn, lang])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_block_calls_equal_the_launch_by_launch_path(ops, dtype, p_drop):
    """The block-level entry points (csrc/blocks.hip: one Python -> C crossing per sublayer and direction) issue the same launches with the
    same kernel choices as the launch-by-launch code: outputs, input gradients and parameter gradients are bit-identical, for one and two
    streams, self-attention (with an additive score bias on stream 0) and FFN, with and without dropout - and the second call of a shape
    really takes the block path."""
    from vln_imagine_amd import _lib
    torch.manual_seed(3)
    B, S0, S1, H, FF = 4, 86, 43, 768, 3072
    mk = lambda *s, sc=0.04: (torch.randn(*s) * sc).cuda().requires_grad_(True)

    def att():
        return (mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1),
                (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True), (0.1 * torch.randn(H)).cuda().requires_grad_(True))

    def ffn():
        return (mk(FF, H), mk(FF, sc=0.1), mk(H, FF, sc=0.02), mk(H, sc=0.1),
                (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True), (0.1 * torch.randn(H)).cuda().requires_grad_(True))

    PA0, PA1, PF0, PF1 = att(), att(), ffn(), ffn()
    x0, x1 = torch.randn(B, S0, H).cuda().to(dtype), torch.randn(B, S1, H).cuda().to(dtype)
    km0 = ((torch.rand(B, S0) < 0.2).float() * -10000.0).cuda()
    km1 = ((torch.rand(B, S1) < 0.2).float() * -10000.0).cuda()
    bias0 = (0.1 * torch.randn(B, S0, S0)).cuda().requires_grad_(True)
    d = lambda seed: (p_drop, p_drop, seed) if p_drop else ops.NO_DROP
    cases = {
        "self-att, one stream": lambda a, b: (ops.self_att_block(a, km0, PA0, drop=d(11)),),
        "ffn, one stream": lambda a, b: (ops.ffn_block(a, PF0, drop=d(21)),),
        "self-att, two streams + bias": lambda a, b: ops.dual_self_att_block(a, b, km0, km1, PA0, PA1, drop0=d(31), drop1=d(41), bias0=bias0),
        "ffn, two streams": lambda a, b: ops.dual_ffn_block(a, b, PF0, PF1, drop0=d(51), drop1=d(61)),
    }
    params = [t for P in (PA0, PA1, PF0, PF1) for t in P] + [bias0]
    saved = ops.BLOCK_CALLS
    counts = {}
    orig = _lib.call

    def counting(name, *a):
        counts[name] = counts.get(name, 0) + 1
        return orig(name, *a)

    def run(fn):
        a, b = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
        for p in params:
            p.grad = None
        outs = fn(a, b)
        torch.manual_seed(5)
        sum((o.float() * torch.randn_like(o.float())).sum() for o in outs).backward()
        return [o.detach().clone() for o in outs], [t.grad.clone() if t.grad is not None else None for t in [a, b] + params]

    try:
        for name, fn in cases.items():
            ops.BLOCK_CALLS = False
            ref = run(fn)
            ops.BLOCK_CALLS = True
            run(fn)                                          # (a shape's first call may time the GEMM pipelines)
            counts.clear()
            _lib.call = counting
            try:
                got = run(fn)
            finally:
                _lib.call = orig
            assert any(k.endswith("_block_fwd") for k in counts) and any(k.endswith("_block_bwd") for k in counts), (name, counts)
            assert not any(k.startswith(("vlni_gemm_nt_dual", "vlni_attn", "vlni_layernorm")) for k in counts), (name, counts)
            for o, r in zip(got[0], ref[0]):
                assert torch.equal(o, r), name
            for gg, gr in zip(got[1], ref[1]):
                assert (gg is None) == (gr is None), name
                if gg is not None:
                    # float atomics (column sums, LayerNorm dgamma / dbeta) add in another order from run to run
                    assert torch.equal(gg, gr) or (gg - gr).abs().max().item() <= 1e-5 * max(1.0, gr.abs().max().item()), name
    finally:
        ops.BLOCK_CALLS = saved
        _lib.call = orig


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_combine_equals_the_unfused_operators(ops, dtype):
    """vlni_embed_combine_fwd - LayerNorm + small-K linear + LayerNorm + gathers + sum-LayerNorm + dropout in ONE launch - against the operators it
    replaces (ops.layer_norm, smallk_linear, sum_layer_norm, the library's counter-based dropout), forward and every gradient, in the four shapes the
    models use: observation embedding (ImageEmbeddings, vilmodel_cmt.py:535-544), history combine with the panorama mean (:611-618), panorama input
    without outer LayerNorm (:603-610), DUET map nodes (no LayerNorm on the image, step-embedding gather, vilmodel.py:1140-1147)."""
    torch.manual_seed(2)
    rows, H, K = 3 * 37, 768, 4
    mk = lambda *s_, sc=1.0: (torch.randn(*s_) * sc).cuda().requires_grad_(True)
    a0 = torch.randn(rows, H).cuda().to(dtype)
    f = torch.randn(rows, K).cuda()
    extra0 = torch.randn(rows, H).cuda().to(dtype)
    idx = torch.randint(0, 3, (rows,)).cuda()
    idx2 = torch.randint(0, 50, (rows,)).cuda()
    ga, ba, gb, beb, go, bo = [(1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True) if i % 2 == 0 else mk(H, sc=0.1) for i in range(6)]
    Wb, bb, row, table, table2 = mk(H, K, sc=0.3), mk(H, sc=0.1), mk(H, sc=0.2), mk(3, H, sc=0.2), mk(50, H, sc=0.2)
    params = [ga, ba, gb, beb, go, bo, Wb, bb, row, table, table2]
    tol = TOL[dtype]

    def unfused(a, extra, kind):
        ta = ops.layer_norm(ops.smallk_linear(f, Wb, bb, dtype), gb, beb, 1e-12)
        if kind == "observation":
            srcs = [(ops.layer_norm(a, ga, ba, 1e-12), "dense", None), (ta, "dense", None), (row, "bcast", None), (table, "gather", idx)]
            return ops.sum_layer_norm(srcs, go, bo, rows, dtype, 1e-12)
        if kind == "history":
            srcs = [(ops.layer_norm(a, ga, ba, 1e-12), "dense", None), (ta, "dense", None), (row, "bcast", None), (extra, "dense", None)]
            return ops.sum_layer_norm(srcs, go, bo, rows, dtype, 1e-12)
        if kind == "panorama input":
            return ops.layer_norm(a, ga, ba, 1e-12) + ta
        return a + table2.index_select(0, idx2).to(dtype) + ta          # map nodes

    def fused(a, extra, kind, p=0.0):
        small = (f, Wb, bb, gb, beb)
        if kind == "observation":
            return ops.embed_combine(a, dtype, ln_a=(ga, ba), small=small, row=row, table=(table, idx), ln_o=(go, bo), p_drop=p, training=True)
        if kind == "history":
            return ops.embed_combine(a, dtype, ln_a=(ga, ba), small=small, row=row, extra=extra, ln_o=(go, bo), p_drop=p, training=True)
        if kind == "panorama input":
            return ops.embed_combine(a, dtype, ln_a=(ga, ba), small=small, p_drop=p, training=True)
        return ops.embed_combine(a, dtype, small=small, table=(table2, idx2), p_drop=p, training=True)

    w = torch.randn(rows, H).cuda()
    for kind in ("observation", "history", "panorama input", "map nodes"):
        res = []
        for fn in (unfused, fused):
            a, extra = a0.clone().requires_grad_(True), extra0.clone().requires_grad_(True)
            for p_ in params:
                p_.grad = None
            y = fn(a, extra, kind)
            (y.float() * w).sum().backward()
            res.append((y.detach().float(), [t.grad.clone().float() if t.grad is not None else None for t in [a, extra] + params]))
        (y0, g0), (y1, g1) = res
        assert (y0 - y1).abs().max().item() <= (1e-5 if dtype == torch.float32 else 2 * tol) * max(1.0, y0.abs().max().item()), kind
        for i, (u, v) in enumerate(zip(g0, g1)):
            assert (u is None) == (v is None), (kind, i)
            if u is not None:
                assert (u - v).abs().max().item() <= (2e-5 if dtype == torch.float32 else 4 * tol) * max(1.0, u.abs().max().item()), (kind, i)
    # dropout: the fused epilogue draws the mask the library's dropout kernel draws for the same seed
    ops.reseed(77)
    y_d = fused(a0, extra0, "observation", p=0.25).float()
    ops.reseed(77)
    seed = ops.next_seeds(1)
    y_ref = ops.dropout_apply(fused(a0, extra0, "observation").contiguous(), 0.25, seed).float()
    assert torch.equal(y_d, y_ref) and float((y_d == 0).float().mean()) > 0.15


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_ln_rowdot_equals_layernorm_dropout_rowdot(ops, dtype, p_drop):
    """vlni_ln_rowdot_fwd (the prediction heads' LayerNorm + dropout + Linear(768 -> 1) + masked_fill in one launch, vilmodel_cmt.py:953-963,1200)
    against ops.layer_norm -> the library's dropout -> ops.row_dot: the same logits (same roundings, same summation order) and gradients."""
    torch.manual_seed(4)
    rows, H = 5 * 37, 768
    x0 = torch.randn(rows, H).cuda().to(dtype)
    g = (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True)
    b = (0.1 * torch.randn(H)).cuda().requires_grad_(True)
    w = (0.05 * torch.randn(1, H)).cuda().requires_grad_(True)
    bias = torch.randn(1).cuda().requires_grad_(True)
    mask = (torch.rand(rows) < 0.3).cuda()
    wt = torch.randn(rows).cuda()
    res = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        for t in (g, b, w, bias):
            t.grad = None
        ops.reseed(99)
        if fused:
            lg = ops.ln_rowdot(x, g, b, w, bias, mask, p_drop=p_drop, training=True)
        else:
            h = ops.layer_norm(x, g, b, 1e-12)
            if p_drop:
                h = ops._TapeDropout.apply(h, p_drop, ops.next_seeds(1))
            lg = ops.row_dot(h, w, bias, mask)
        fin = torch.isfinite(lg)
        assert torch.equal(~fin, mask)
        (lg[fin] * wt[fin]).sum().backward()
        res.append((lg.detach().clone(), [t.grad.clone() for t in (x, g, b, w, bias)]))
    (l0, g0), (l1, g1) = res
    fin = torch.isfinite(l0)
    # same arithmetic, but the compiler contracts the two kernels' multiply-adds differently: last-bit differences in float32, and a rare
    # flipped rounding of one of the 768 activation-dtype elements of a row in bf16
    tol = 2e-6 if dtype == torch.float32 else 1e-3
    assert (l0[fin] - l1[fin]).abs().max().item() <= tol * l0[fin].abs().max().item()
    for u, v in zip(g0, g1):
        assert (u.float() - v.float()).abs().max().item() <= max(tol, 1e-5) * max(1.0, u.float().abs().max().item())


def test_launch_table_uploaded_inside_a_capture(ops):
    """ops._dev_table while a stream is capturing: the table goes through the pinned staging buffer as a memcpy node of the graph (torch's own
    host -> device copy is not capturable), so a captured step whose reduction table no eager step has built yet still captures and replays."""
    import numpy as np
    ops.reserve_staging()
    n = 4096
    dst = torch.zeros(n, device="cuda")
    parts = torch.arange(3 * n, device="cuda", dtype=torch.float32).reshape(3, n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g):
            arr = np.zeros((1,), ops._PART_DT)
            arr[0] = (dst.data_ptr(), parts.data_ptr(), n // 4, n // 4, 3 | ops.PART_STORE, 0)
            tab = ops._dev_table(arr, dst.device)
            ops._lib.call("vlni_reduce_parts_sq", tab.data_ptr(), 1, (n // 4 + 1023) // 1024, None, 1, ops._st())
    want = parts.sum(0)
    for _ in range(2):
        dst.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(dst, want)


def test_reduce_parts_modes_and_spread_sum_of_squares(ops):
    """vlni_reduce_parts_sq entry modes (include/vlni.h VLNI_PART_*): add onto dst, store, zero fill (store with no partials), sum of squares only
    (no write), and the sum of squares of what each flagged entry leaves in dst, spread over the slots and folded by vlni_sumsq_fold."""
    import numpy as np
    torch.manual_seed(2)
    n, sp = 8192 + 1024 + 4, 3                                  # not a multiple of a block's 4096 floats
    parts = torch.randn(sp, n, device="cuda")
    mk = lambda: torch.randn(n, device="cuda")
    d_add, d_store, d_zero, d_ro = mk(), mk(), mk(), mk()
    want = {"add": d_add + parts.sum(0), "store": parts[0] + parts[1] + parts[2], "zero": torch.zeros(n, device="cuda"), "ro": d_ro.clone()}
    ents = [(d_add, parts, sp | ops.PART_SUMSQ), (d_store, parts, sp | ops.PART_STORE | ops.PART_SUMSQ), (d_zero, None, ops.PART_STORE),
            (d_ro, None, ops.PART_NOWRITE | ops.PART_SUMSQ)]
    arr, blk = np.zeros((len(ents),), ops._PART_DT), 0
    for i, (dst, prt, flags) in enumerate(ents):
        arr[i] = (dst.data_ptr(), prt.data_ptr() if prt is not None else 0, n // 4, n // 4, flags, blk)
        blk += -(-(n // 4) // 1024)
    tab = ops._dev_table(arr, d_add.device)
    slots = torch.zeros(32 * ops.SUMSQ_SLOTS, device="cuda")
    total = torch.zeros(1, device="cuda")
    ops._lib.call("vlni_reduce_parts_sq", tab.data_ptr(), len(ents), blk, slots.data_ptr(), ops.SUMSQ_SLOTS, ops._st())
    ops._lib.call("vlni_sumsq_fold", slots.data_ptr(), ops.SUMSQ_SLOTS, total.data_ptr(), ops._st())
    for got, key in ((d_add, "add"), (d_store, "store"), (d_zero, "zero"), (d_ro, "ro")):
        assert (got - want[key]).abs().max().item() <= 1e-5, key
    ss = sum(float((want[k].double() ** 2).sum()) for k in ("add", "store", "ro"))
    assert abs(total.item() - ss) <= 1e-5 * ss
    assert int((slots.view(-1, 32)[:, 0] != 0).sum()) >= min(blk, ops.SUMSQ_SLOTS) // 2 and float(slots.view(-1, 32)[:, 1:].abs().sum()) == 0.0


def test_stale_weight_copies_are_recast_in_one_launch(ops):
    """ShadowCache._refresh_plain / vlni_shadow_refresh: after an in-place update of plain float32 parameters (what torch.optim's step is to the
    cache) every cached copy - single, row-packed Q | K | V, transposed, bias vectors, and (round 6) the float32 row-pack of the Q | K | V biases
    that the GEMM epilogue reads - is rewritten in place by ONE launch per destination type and equals a fresh cast; the tensors handed out
    before stay the same objects (captured graphs hold their addresses, vln_imagine_amd/graphed.py)."""
    torch.manual_seed(3)
    mkp = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda"))
    q, k, v, o = mkp(768, 768), mkp(768, 768), mkp(768, 768), mkp(768, 3072)
    bq, bk, bv = mkp(768), mkp(768), mkp(768)
    dt = torch.bfloat16
    was = ops.BATCH_SHADOWS
    ops.BATCH_SHADOWS = True
    calls, real = [], ops._lib.call
    ops.SHADOWS._c.clear()                     # (copies of other tests' parameters would be refreshed by the same launches)
    try:
        get = lambda: (ops.SHADOWS.get((q, k, v), dt), ops.SHADOWS.get((q, k, v), dt, True), ops.SHADOWS.get((o,), dt), ops.SHADOWS.get((o,), dt, True),
                       ops.SHADOWS.get((bq, bk, bv), dt), ops.SHADOWS.get((bq, bk, bv), torch.float32))
        first = get()
        with torch.no_grad():
            for p in (q, k, v, o, bq, bk, bv):
                p.mul_(1.5).add_(0.25)
        ops._lib.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        second = get()
    finally:
        ops._lib.call = real
        ops.BATCH_SHADOWS = was
    assert calls.count("vlni_shadow_refresh") == 2 and not any(c in ("vlni_cast", "vlni_transpose") for c in calls), calls      # bf16 + float32
    assert all(a is b for a, b in zip(first, second))
    W = torch.cat([q, k, v], 0).detach()
    assert torch.equal(second[0], W.to(dt)) and torch.equal(second[1], W.to(dt).t().contiguous())
    assert torch.equal(second[2], o.detach().to(dt)) and torch.equal(second[3], o.detach().to(dt).t().contiguous())
    assert torch.equal(second[4], torch.cat([bq, bk, bv]).detach().to(dt))
    assert second[5].dtype == torch.float32 and torch.equal(second[5], torch.cat([bq, bk, bv]).detach())
    assert ops.SHADOWS.current_for_replay()


def test_small_ops(ops):
    import torch.nn.functional as F
    torch.manual_seed(0)
    dev = "cuda"
    # cross entropy with -inf logits and ignore_index
    lg = torch.randn(6, 37, device=dev)
    lg[:, 30:] = -float("inf")
    tgt = torch.tensor([0, 5, -100, 29, 3, -100], device=dev)
    lr = lg.clone().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt, ignore_index=-100, reduction="sum")
    (ref * 0.3).backward()
    lp = lg.clone().requires_grad_(True)
    out = ops.cross_entropy_sum(lp, tgt)
    (out * 0.3).backward()
    assert _err(out, ref) < 1e-5 and _err(lp.grad, lr.grad) < 1e-6
    # row dot with mask
    h = torch.randn(4, 9, 768, device=dev, requires_grad=True)
    w = torch.randn(1, 768, device=dev, requires_grad=True)
    bb = torch.randn(1, device=dev, requires_grad=True)
    mask = torch.rand(4, 9, device=dev) > 0.7
    o = ops.row_dot(h, w, bb, mask)
    ref = (h.double() @ w.double().t()).squeeze(-1) + bb.double()
    assert torch.isinf(o[mask]).all() and _err(o[~mask], ref[~mask]) < 1e-4
    wts = torch.randn(4, 9, device=dev)
    (o.masked_fill(mask, 0) * wts).sum().backward()
    gh, gw, gb = h.grad.clone(), w.grad.clone(), bb.grad.clone()
    h.grad = w.grad = bb.grad = None
    (ref.masked_fill(mask, 0) * wts.double()).sum().backward()
    assert _err(gh, h.grad) < 1e-5 and _err(gw, w.grad) < 1e-4 and _err(gb, bb.grad) < 1e-5
    # small-K linear
    x = torch.randn(50, 4, device=dev)
    W = torch.randn(768, 4, device=dev, requires_grad=True)
    b2 = torch.randn(768, device=dev, requires_grad=True)
    y = ops.smallk_linear(x, W, b2, torch.float32)
    yr = F.linear(x.double(), W.double(), b2.double())
    assert _err(y, yr) < 1e-5
    gy = torch.randn_like(y)
    y.backward(gy)
    gW, gb2 = W.grad.clone(), b2.grad.clone()
    W.grad = b2.grad = None
    yr.backward(gy.double())
    assert _err(gW, W.grad) < 1e-4 and _err(gb2, b2.grad) < 1e-4
    # seq mean, cosine, segment mean
    xs = torch.randn(5, 36, 768, device=dev, requires_grad=True)
    m = ops.seq_mean(xs)
    assert _err(m, xs.double().mean(1)) < 1e-6
    m.sum().backward()
    assert _err(xs.grad, torch.full_like(xs, 1 / 36)) < 1e-7
    a = torch.randn(7, 768, device=dev, requires_grad=True)
    c = torch.randn(7, 768, device=dev, requires_grad=True)
    cs = ops.cosine(a, c)
    ar, cr = a.double().detach().requires_grad_(True), c.double().detach().requires_grad_(True)
    csr = F.cosine_similarity(ar, cr, dim=-1)
    assert _err(cs, csr) < 1e-6
    ww = torch.randn(7, device=dev)
    (cs * ww).sum().backward(); (csr * ww.double()).sum().backward()
    assert _err(a.grad, ar.grad) < 1e-6 and _err(c.grad, cr.grad) < 1e-6
    t = torch.randn(40, 768, device=dev, requires_grad=True)
    off = torch.tensor([0, 3, 4, 9], dtype=torch.int32, device=dev)
    rows = torch.tensor([1, 2, 3, 10, 20, 21, 22, 3, 39], dtype=torch.int32, device=dev)
    sm = ops.segment_mean(t, off, rows)
    tr = t.double().detach().requires_grad_(True)
    smr = torch.stack([tr[[1, 2, 3]].mean(0), tr[[10]].mean(0), tr[[20, 21, 22, 3, 39]].mean(0)])
    assert _err(sm, smr) < 1e-6
    wz = torch.randn_like(sm)
    (sm * wz).sum().backward(); (smr * wz.double()).sum().backward()
    assert _err(t.grad, tr.grad) < 1e-6
    # sum + layernorm with gather / bcast / dense sources
    tab = torch.randn(11, 768, device=dev, requires_grad=True)
    row = torch.randn(768, device=dev, requires_grad=True)
    dn = torch.randn(30, 768, device=dev, requires_grad=True)
    idx = torch.randint(0, 11, (30,), device=dev)
    g = (1 + 0.1 * torch.randn(768, device=dev)).requires_grad_(True)
    be = (0.1 * torch.randn(768, device=dev)).requires_grad_(True)
    y = ops.sum_layer_norm([(tab, "gather", idx), (row, "bcast", None), (dn, "dense", None)], g, be, 30, torch.float32)
    leaves = [tab, row, dn, g, be]
    dl = [l.double().detach().requires_grad_(True) for l in leaves]
    yr = F.layer_norm(dl[0][idx] + dl[1][None] + dl[2], (768,), dl[3], dl[4], 1e-12)
    assert _err(y, yr) < 1e-5
    wy = torch.randn_like(y)
    (y * wy).sum().backward(); (yr * wy.double()).sum().backward()
    for a_, r_ in zip(leaves, dl):
        assert _err(a_.grad, r_.grad) < 1e-4 * max(1.0, r_.grad.abs().max().item())


def test_adamw_and_clip(ops):
    from vln_imagine_amd import _lib
    torch.manual_seed(0)
    n = 4096 * 3
    p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 3
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for step in (1, 2, 3):
        pr.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_([pr], 40.0)
        opt.step()
        ss = torch.zeros(1, device="cuda"); coef = torch.empty(1, device="cuda")
        _lib.call("vlni_sumsq", g.data_ptr(), n, ss.data_ptr(), st)
        _lib.call("vlni_clip_coef", ss.data_ptr(), 40.0, coef.data_ptr(), st)
        assert abs(math.sqrt(ss.item()) - total.item()) < 1e-2
        _lib.call("vlni_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, 1e-3, 0.9, 0.999,
                  1e-8, 0.01, step, coef.data_ptr(), st)
        assert _err(p, pr.detach()) < 1e-5, step
    assert _err(sh.float(), p) < 2e-2


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (148, 768, 768), (5504, 768, 768), (333, 3072, 768), (70, 768, 3072), (2431, 1536, 768)])
def test_wgrad_tn_bf16_exact_and_random(ops, M, N, K):
    """transposing-read TN kernel: exact on small integers (catches any tr-read lane/row slip), close on random data,
    fused bias gradient, accumulation into an existing buffer."""
    g = torch.Generator().manual_seed(7)
    dyi = torch.randint(-2, 3, (M, N), generator=g).float()
    xi = torch.randint(-2, 3, (M, K), generator=g).float()
    out, cs = ops.wgrad(dyi.bfloat16().cuda(), xi.bfloat16().cuda(), want_colsum=True)
    assert torch.equal(out.cpu(), dyi.t() @ xi), "integer wgrad mismatch"
    assert torch.equal(cs.cpu(), dyi.sum(0))
    dy, x = _rand((M, N), torch.bfloat16, 5, 0.1), _rand((M, K), torch.bfloat16, 6, 0.5)
    ref = dy.double().t() @ x.double()
    out = ops.wgrad(dy, x)
    assert _err(out, ref) < 2e-3 * max(1.0, ref.abs().max().item())
    out2, cs2 = ops.wgrad(dy, x, out=out, want_colsum=True)
    assert _err(out2, 2 * ref) < 4e-3 * max(1.0, ref.abs().max().item())
    assert _err(cs2, dy.double().sum(0)) < 2e-3 * max(1.0, dy.double().sum(0).abs().max().item())


@pytest.mark.parametrize("split", [1, 3, 5])
@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8])
def test_wgrad_tn_lds_dma_variants_exact(ops, variant, split):
    """LDS-DMA pipelines of the grouped transposing-read wgrad (zero page for row tails, MFMA-ones bias gradient): exact on
    small integers over several ragged segments."""
    import ctypes
    from vln_imagine_amd import _lib
    g = torch.Generator().manual_seed(11)
    N, K, Ms = 768, 256, [700, 64, 333]
    dys = [torch.randint(-2, 3, (m, N), generator=g).float() for m in Ms]
    xs = [torch.randint(-2, 3, (m, K), generator=g).float() for m in Ms]
    dd, xx = [d.bfloat16().cuda() for d in dys], [x.bfloat16().cuda() for x in xs]
    out, cs = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    n = len(Ms)
    pa = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dd]); pb = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xx])
    pm = (ctypes.c_int * n)(*Ms)
    _lib.call("vlni_gemm_tn_bf16_grouped_v", n, pa, pb, pm, N, K, out.data_ptr(), K, N, K, cs.data_ptr(), split, variant,
              torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out.cpu(), sum(d.t() @ x for d, x in zip(dys, xs)))
    assert torch.equal(cs.cpu(), sum(d.sum(0) for d in dys))


@pytest.mark.parametrize("split", [2, 4])
@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8])
def test_wgrad_partials_then_batched_reduction_exact(ops, variant, split):
    """Partials mode: every row split stores its share (no atomics), vlni_reduce_parts adds all of them into gradients that already
    hold something - two tensors (weight + bias gradient) x two parameters in ONE reduction launch. Exact on small integers."""
    import ctypes
    import numpy as np
    from vln_imagine_amd import _lib
    g = torch.Generator().manual_seed(13 + variant)
    st = torch.cuda.current_stream().cuda_stream
    entries, refs, keep = [], [], []
    for (N, K, Ms) in ((768, 256, [700, 64, 333]), (256, 776, [512, 130])):
        dys = [torch.randint(-2, 3, (m, N), generator=g).float() for m in Ms]
        xs = [torch.randint(-2, 3, (m, K), generator=g).float() for m in Ms]
        dd, xx = [d.bfloat16().cuda() for d in dys], [x.bfloat16().cuda() for x in xs]
        n = len(Ms)
        nmt = sum((m + 63) // 64 for m in Ms)
        eff, per = ops._eff_split(nmt, split)
        assert per >= 3
        part = torch.full((eff * (N * K + N),), 7.0, device="cuda")                   # stale values must all be overwritten
        w0, b0 = torch.randint(-3, 4, (N, K), generator=g).float(), torch.randint(-3, 4, (N,), generator=g).float()
        gw, gb = w0.cuda(), b0.cuda()
        pa = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dd]); pb = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xx])
        pm = (ctypes.c_int * n)(*Ms)
        cpart = part.data_ptr() + 4 * eff * N * K
        _lib.call("vlni_gemm_tn_bf16_grouped_part", n, pa, pb, pm, N, K, part.data_ptr(), N * K, N, K, cpart, split, variant, st)
        entries += [(gw.data_ptr(), part.data_ptr(), N * K // 4, N * K // 4, eff), (gb.data_ptr(), cpart, N // 4, N // 4, eff)]
        refs.append((gw, w0 + sum(d.t() @ x for d, x in zip(dys, xs)), gb, b0 + sum(d.sum(0) for d in dys)))
        keep += [dd, xx, part]
    arr = np.zeros((len(entries),), ops._PART_DT)
    blk = 0
    for i, (dst, part_ptr, n4, s4, eff) in enumerate(entries):
        arr[i] = (dst, part_ptr, n4, s4, eff, blk)
        blk += -(-n4 // 1024)
    tab = torch.from_numpy(arr.view(np.uint8)).cuda()
    _lib.call("vlni_reduce_parts", tab.data_ptr(), len(entries), blk, st)
    for gw, rw, gb, rb in refs:
        assert torch.equal(gw.cpu(), rw) and torch.equal(gb.cpu(), rb)


def test_wgrad_partials_reject_register_staged_kernel(ops):
    import ctypes
    from vln_imagine_amd import _lib
    d, x = torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda"), torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(2 * (64 * 64 + 64), device="cuda")
    pa, pb, pm = (ctypes.c_void_p * 1)(d.data_ptr()), (ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_int * 1)(128)
    with pytest.raises(_lib.VlniError):                                              # 1 row tile per split: no LDS-DMA pipeline
        _lib.call("vlni_gemm_tn_bf16_grouped_part", 1, pa, pb, pm, 64, 64, part.data_ptr(), 64 * 64, 64, 64, part.data_ptr() + 4 * 2 * 4096,
                  2, 5, torch.cuda.current_stream().cuda_stream)


def _same(x, y, variant):
    """Every pipeline on v_mfma_32x32x16 sums the same products in the same tree: equal bit for bit. Variants 15 and 32 (the 256 x 256 and
    256 x 128 kernels on v_mfma_16x16x32) sum them in another tree: equal to float32 rounding of the accumulator, i.e. <= 1 ulp of the 16-bit output
    on a small fraction of the elements (float32 operands never reach it, they run variant 14)."""
    if variant not in (15, 32) or x.dtype == torch.float32:
        return torch.equal(x, y)
    xf, yf = x.float(), y.float()
    ulp = 2.0 ** -7 if x.dtype == torch.bfloat16 else 2.0 ** -10
    close = bool(((xf - yf).abs() <= ulp * torch.maximum(xf.abs(), yf.abs()) + 1e-6).all())
    return close and float((xf != yf).float().mean()) < 0.05


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 32])
def test_gemm_variants_identical(ops, variant):
    """All GEMM pipelines (register-staged, LDS-DMA 2/3-stage with 4 or 8 waves, large tiles 256x128 / 256x256 / 128x256, persistent)
    give bit-identical results, incl. epilogues and ragged tile edges; the 8-phase kernel (15) agrees to accumulator rounding."""
    for dtype in (torch.bfloat16, torch.float32):
        for (M, N, K) in ((333, 768, 768), (700, 640, 384)):
            a, b = _rand((M, K), dtype, 31, 0.5), _rand((N, K), dtype, 32, 0.05)
            bias, res = _rand((N,), torch.float32, 33, 0.1), _rand((M, N), dtype, 34, 0.5)
            outs = []
            for v in (1, variant):
                out, z = torch.empty((M, N), dtype=dtype, device="cuda"), torch.empty((M, N), dtype=dtype, device="cuda")
                ops._gemm_call(v, a, b, out, bias, 1, res, z, None, 0, 1.0, 1, False, M, N, K)
                outs.append((out, z))
            assert _same(outs[0][0], outs[1][0], variant) and _same(outs[0][1], outs[1][1], variant), (variant, dtype, M, N, K)


@pytest.mark.parametrize("variant", [1, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 32])
def test_gemm_dual_launch_variants(ops, variant):
    """Two problems in one launch (language + vision stream) == two single launches, for every tile geometry."""
    saved = (ops.AUTOTUNE, ops.GEMM_VARIANTS)
    try:
        ops.AUTOTUNE, ops.GEMM_VARIANTS = True, (variant,)
        ops._GEMM_BEST.clear()
        for dtype in (torch.bfloat16, torch.float32):
            N, K = 768, 768
            a0, a1 = _rand((5 * 86, K), dtype, 41, 0.5), _rand((5 * 44 + 3, K), dtype, 42, 0.5)
            b0, b1 = _rand((N, K), dtype, 43, 0.05), _rand((N, K), dtype, 44, 0.05)
            bias = (_rand((N,), torch.float32, 45, 0.1), _rand((N,), torch.float32, 46, 0.1))
            res = (_rand((a0.shape[0], N), dtype, 47, 0.5), _rand((a1.shape[0], N), dtype, 48, 0.5))
            o0, o1 = ops.gemm_nt2((a0, a1), (b0, b1), bias=bias, residual=res)
            r0 = torch.empty_like(o0); r1 = torch.empty_like(o1)
            ops._gemm_call(1, a0, b0, r0, bias[0], 0, res[0], None, None, 0, 1.0, 1, False, a0.shape[0], N, K)
            ops._gemm_call(1, a1, b1, r1, bias[1], 0, res[1], None, None, 0, 1.0, 1, False, a1.shape[0], N, K)
            assert _same(o0, r0, variant) and _same(o1, r1, variant), (variant, dtype)
    finally:
        ops.AUTOTUNE, ops.GEMM_VARIANTS = saved
        ops._GEMM_BEST.clear()


@pytest.mark.parametrize("nprob", [1, 3, 4])
@pytest.mark.parametrize("variant", [1, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 32])
def test_gemm_multi_problem_launch_variants(ops, variant, nprob):
    """vlni_gemm_nt_multi: 1, 3 or 4 problems of one (N, K, epilogue kind) in ONE launch (the language / vision streams of a cross-modal layer
    + the history panorama encoder's layer; ragged row counts so that every problem ends in a partial tile) == single launches of the
    register-staged kernel, for every tile geometry, with the epilogues a step uses (bias + residual + dropout; GELU with stored
    pre-activation); and against float64 directly."""
    saved = (ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS)
    rows = (5 * 86, 5 * 44 + 3, 5 * 36, 77)[:nprob]
    try:
        ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = True, (variant,), 1 << 30, 1 << 30
        for dtype in (torch.bfloat16, torch.float32):
            for (N, K) in ((768, 768), (640, 384)):
                ops._GEMM_BEST.clear()
                a = [_rand((m, K), dtype, 141 + i, 0.5) for i, m in enumerate(rows)]
                b = [_rand((N, K), dtype, 151 + i, 0.05) for i in range(nprob)]
                bias = [_rand((N,), torch.float32, 161 + i, 0.1) for i in range(nprob)]
                res = [_rand((m, N), dtype, 171 + i, 0.5) for i, m in enumerate(rows)]
                seeds = tuple(1000 + 17 * i for i in range(nprob))
                outs = ops.gemm_ntn(a, b, bias=bias, residual=res, drop=(0.25, seeds))
                z = [torch.empty_like(r) for r in res]
                acts = ops.gemm_ntn(a, b, bias=bias, act=1, preact=z)
                for i in range(nprob):
                    r = torch.empty_like(outs[i]); rz = torch.empty_like(outs[i]); ra = torch.empty_like(outs[i])
                    ops._gemm_call(1, a[i], b[i], r, bias[i], 0, res[i], None, None, 0, 1.0, 1, False, rows[i], N, K, drop=(0.25, seeds[i]))
                    ops._gemm_call(1, a[i], b[i], ra, bias[i], 1, None, rz, None, 0, 1.0, 1, False, rows[i], N, K)
                    assert _same(outs[i], r, variant), (variant, dtype, N, K, i, "residual + dropout")
                    assert _same(acts[i], ra, variant) and _same(z[i], rz, variant), (variant, dtype, N, K, i, "gelu + pre-activation")
                    pre = a[i].double() @ b[i].double().t() + bias[i].double()
                    _chk(z[i], pre, TOL[dtype] if dtype in TOL else 1e-5, (variant, dtype, N, K, i, "float64"))
    finally:
        ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = saved
        ops._GEMM_BEST.clear()


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6])
def test_gemm_nn_weight_layout_matches_transposed_copy(ops, variant):
    """dgrad straight from W[out, in] (transposing LDS reads, variant + 16) == the NT kernel on an explicit W^T copy, bit for bit,
    incl. GELU' / residual epilogues, ragged rows and a partial column tile (zero page)."""
    for (M, N, K) in ((333, 768, 768), (700, 640, 3072), (512, 3072, 768)):
        a = _rand((M, K), torch.bfloat16, 51, 0.5)
        w = _rand((K, N), torch.bfloat16, 52, 0.05)                      # [out = K, in = N]
        wt = w.t().contiguous()
        res, z = _rand((M, N), torch.bfloat16, 53, 0.5), _rand((M, N), torch.bfloat16, 54, 1.0)
        for kw in (dict(), dict(residual=res), dict(dact_src=z, dact=1)):
            r = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            o = torch.empty_like(r)
            ops._gemm_call(1, a, wt, r, None, 0, kw.get("residual"), None, kw.get("dact_src"), kw.get("dact", 0), 1.0, 1, False, M, N, K)
            ops._gemm_call(16 + variant, a, w, o, None, 0, kw.get("residual"), None, kw.get("dact_src"), kw.get("dact", 0), 1.0, 1, False,
                           M, N, K)
            assert torch.equal(o, r), (variant, M, N, K, list(kw))
    # front-end: KN operands through gemm_nt / gemm_nt2 (dual launch)
    a0, a1 = _rand((5 * 86, 768), torch.bfloat16, 55, 0.5), _rand((5 * 44 + 3, 768), torch.bfloat16, 56, 0.5)
    w0, w1 = _rand((768, 2304), torch.bfloat16, 57, 0.05), _rand((768, 2304), torch.bfloat16, 58, 0.05)
    o0, o1 = ops.gemm_nt2((a0, a1), (ops.KN(w0), ops.KN(w1)))
    assert torch.equal(o0, ops.gemm_nt(a0, w0.t().contiguous())) and torch.equal(o1, ops.gemm_nt(a1, ops.KN(w1)))


@pytest.mark.parametrize("nn,pk", [(False, 14), (True, 6), (False, 15), (False, 32)])
def test_gemm_persistent_kernel_streams_many_tiles(ops, nn, pk):
    """The persistent kernel (variant 14; 16 + 6 with the weight as [K, N]) at the bench's row counts: > 512 tiles, so every block walks
    several tiles through its two LDS stages (next tile's first k-tile prefetched under the epilogue), two problems per launch, every
    epilogue kind (bias + GELU + stored pre-activation, GELU' + residual, dropout), ragged last row tile. Bit-identical to variant 1 / 5
    (15, the 256 x 256 8-phase kernel: to accumulator rounding; its k-tile stream runs on across output tiles of BOTH problems)."""
    base = 16 if nn else 0
    ref_v, pk_v = (16 + 5, 16 + pk) if nn else (1, pk)
    K = 768
    for N in (768, 2304):
        M0, M1 = 5504, 2752 + 37
        a0, a1 = _rand((M0, K), torch.bfloat16, 61, 0.5), _rand((M1, K), torch.bfloat16, 62, 0.5)
        w0, w1 = _rand((N, K), torch.bfloat16, 63, 0.05), _rand((N, K), torch.bfloat16, 64, 0.05)
        if nn:                                          # operands of the dgrad form: B given as [K, N]
            w0, w1 = w0.t().contiguous(), w1.t().contiguous()
        bias = (_rand((N,), torch.float32, 65, 0.1), _rand((N,), torch.float32, 66, 0.1))
        res = (_rand((M0, N), torch.bfloat16, 67, 0.5), _rand((M1, N), torch.bfloat16, 68, 0.5))
        zsrc = (_rand((M0, N), torch.bfloat16, 69, 1.0), _rand((M1, N), torch.bfloat16, 70, 1.0))
        for kind in ("gelu_preact", "dact_res", "drop_res"):
            outs = []
            for v in (ref_v, pk_v):
                saved = (ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.NN_VARIANTS)
                try:
                    ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.NN_VARIANTS = True, (v - base,), (v - base,)
                    ops._GEMM_BEST.clear()
                    b = (ops.KN(w0), ops.KN(w1)) if nn else (w0, w1)
                    if kind == "gelu_preact":
                        z = (torch.empty((M0, N), dtype=torch.bfloat16, device="cuda"), torch.empty((M1, N), dtype=torch.bfloat16, device="cuda"))
                        o = ops.gemm_nt2((a0, a1), b, bias=bias, act=1, preact=z)
                        outs.append(o + z)
                    elif kind == "dact_res":
                        outs.append(ops.gemm_nt2((a0, a1), b, dact_src=zsrc, dact=1, residual=res))
                    else:
                        outs.append(ops.gemm_nt2((a0, a1), b, bias=bias, residual=res, drop=(0.1, (1234, 777))))
                    # and the single-problem entry point on the longer stream
                    bs = ops.KN(w0) if nn else w0
                    outs[-1] = tuple(outs[-1]) + (ops.gemm_nt(a0, bs, bias=bias[0], residual=res[0]),)
                finally:
                    ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.NN_VARIANTS = saved
                    ops._GEMM_BEST.clear()
            for x, y in zip(outs[0], outs[1]):
                assert _same(x, y, pk), (nn, N, kind)


@pytest.mark.parametrize("with_bias,p_drop", [(False, 0.0), (True, 0.0), (False, 0.1), (True, 0.1)])
def test_attention_dual_launch_equals_two_launches(ops, with_bias, p_drop):
    """vlni_attn_fwd_dual / _bwd_dual (two problems per launch: the two directions of the bidirectional cross-attention, or the two
    streams' self-attention with DUET's graph bias on the first) == two single launches, bit for bit, on strided q/k/v views."""
    B, nh, H = 5, 12, 768
    for (S0, K0, S1, K1) in ((86, 38, 38, 86), (86, 86, 41, 41), (20, 130, 33, 7)):
        g = torch.Generator().manual_seed(7)
        mk = lambda rows, cols: (torch.randn(rows, cols, generator=g) * 0.7).to(torch.bfloat16).cuda()
        qa, ka = mk(B * S0, 3 * H), mk(B * K0, 3 * H)
        qb, kb = mk(B * S1, 3 * H), mk(B * K1, 3 * H)
        q = (qa[:, :H], qb[:, :H]); k = (ka[:, H:2 * H], kb[:, H:2 * H]); v = (ka[:, 2 * H:], kb[:, 2 * H:])
        km = ((torch.rand(B, K0, generator=g) < 0.2).float().cuda() * -10000.0, (torch.rand(B, K1, generator=g) < 0.2).float().cuda() * -10000.0)
        bias0 = (torch.randn(B, S0, K0, generator=g) * 0.3).cuda() if with_bias else None
        drop1 = lambda i: (p_drop, 100 + i)
        ref = [ops.attn_fwd(q[i], k[i], v[i], B, (S0, S1)[i], (K0, K1)[i], km[i], bias0 if i == 0 else None, nh, drop=drop1(i)) for i in range(2)]
        (o0, l0), (o1, l1) = ops.attn_fwd2(q, k, v, B, (S0, S1), (K0, K1), km, bias0, nh, drop=(p_drop, (100, 101)))
        assert torch.equal(o0, ref[0][0]) and torch.equal(o1, ref[1][0]) and torch.equal(l0, ref[0][1]) and torch.equal(l1, ref[1][1])
        do = (mk(B * S0, H), mk(B * S1, H))
        grads = []
        for dual in (False, True):
            dqa, dka = torch.zeros_like(qa), torch.zeros_like(ka)
            dqb, dkb = torch.zeros_like(qb), torch.zeros_like(kb)
            dq = (dqa[:, :H], dqb[:, :H]); dk = (dka[:, H:2 * H], dkb[:, H:2 * H]); dv = (dka[:, 2 * H:], dkb[:, 2 * H:])
            db = torch.zeros_like(bias0) if with_bias else None
            if dual:
                ops.attn_bwd2(q, k, v, (o0, o1), do, (l0, l1), dq, dk, dv, B, (S0, S1), (K0, K1), km, bias0, db, nh, drop=(p_drop, (100, 101)))
            else:
                for i in range(2):
                    ops.attn_bwd(q[i], k[i], v[i], (o0, o1)[i], do[i], (l0, l1)[i], dq[i], dk[i], dv[i], B, (S0, S1)[i], (K0, K1)[i], km[i],
                                 bias0 if i == 0 else None, db if i == 0 else None, nh, drop=drop1(i))
            grads.append((dqa, dka, dqb, dkb, db))
        for x, y in zip(*grads):
            if x is None:
                continue
            if x.dtype == torch.float32:            # dbias: float atomics over heads, order-dependent in the last bits
                assert (x - y).abs().max().item() <= 1e-5 * max(1.0, y.abs().max().item())
            elif max(S0, K0, S1, K1) <= 96 or min(max(S0, K0), max(S1, K1)) > 96:
                assert torch.equal(x, y)            # both launch forms run the same kernel body
            else:                                   # one problem beyond 96 rows: the dual launch runs both on the chunked kernel, the single
                assert (x.float() - y.float()).abs().max().item() <= 2.0 ** -6 * y.float().abs().max().item()   # launches split (roles / chunked)


def test_float16_kernels_are_the_bfloat16_kernels_with_the_f16_mfma(ops):
    """float16 (BASELINE.json configs[4]) runs csrc/*_impl.inc compiled a second time with _Float16 / v_mfma_f32_32x32x16_f16: every
    GEMM pipeline (incl. the persistent and the [K,N] dgrad kernels), the dual launches, the grouped weight gradient and the fused
    attention give the float64 answer within float16 rounding, and all variants agree bit for bit like their bfloat16 twins."""
    dt = torch.float16
    M, N, K = 700, 768, 768
    a, w = _rand((M, K), dt, 91, 0.5), _rand((N, K), dt, 92, 0.05)
    bias, res = _rand((N,), torch.float32, 93, 0.1), _rand((M, N), dt, 94, 0.5)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double()) + res.double()
    outs = []
    for v in (1, 5, 13, 14):
        out, z = torch.empty((M, N), dtype=dt, device="cuda"), torch.empty((M, N), dtype=dt, device="cuda")
        ops._gemm_call(v, a, w, out, bias, 1, res, z, None, 0, 1.0, 1, False, M, N, K)
        assert _err(out, ref) < 8e-3 * 3
        outs.append(out)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    wt = w.t().contiguous()                                   # [K, N] operand of the transposing-read kernels
    o_nn = torch.empty((M, N), dtype=dt, device="cuda")
    o_nt = torch.empty((M, N), dtype=dt, device="cuda")
    ops._gemm_call(16 + 6, a, wt, o_nn, None, 0, res, None, None, 0, 1.0, 1, False, M, N, K)
    ops._gemm_call(1, a, w, o_nt, None, 0, res, None, None, 0, 1.0, 1, False, M, N, K)
    assert torch.equal(o_nn, o_nt)
    dy, x = _rand((2048, 768), dt, 95, 0.1), _rand((2048, 768), dt, 96, 0.5)
    g = ops.wgrad(dy, x)
    rg = dy.double().t() @ x.double()
    assert _err(g, rg) < 2e-3 * max(1.0, rg.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Sl", [86, 1])
def test_gate_rows_matches_torch(ops, dtype, Sl):
    """f = visn[:, r0:r0+n] * lang[:, :1] (vilmodel_cmt.py:1192) and both full-size gradients, against torch in float64."""
    B, Sv, H, r0, n = 5, 23, 768, 4, 17
    visn, lang = _rand((B, Sv, H), dtype, 71, 1.0).requires_grad_(), _rand((B, Sl, H), dtype, 72, 1.0).requires_grad_()
    df = _rand((B, n, H), dtype, 73, 1.0)
    f = ops.gate_rows(visn, lang, r0, n)
    f.backward(df)
    v64, l64 = visn.detach().double().requires_grad_(), lang.detach().double().requires_grad_()
    ref = v64[:, r0:r0 + n] * l64[:, :1]
    ref.backward(df.double())
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert (f.double() - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert (visn.grad.double() - v64.grad).abs().max().item() <= tol * v64.grad.abs().max().item()
    assert (lang.grad.double() - l64.grad).abs().max().item() <= tol * l64.grad.abs().max().item()
    assert float(visn.grad[:, :r0].abs().max()) == 0.0 and float(visn.grad[:, r0 + n:].abs().max()) == 0.0
    if Sl > 1:
        assert float(lang.grad[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_dual_layernorm_equals_two_launches(ops, dtype, p_drop):
    """vlni_layernorm_fwd_dual / _bwd_dual (the two streams of a cross-modal layer in one launch) == the single-problem launches:
    outputs, statistics, dx and the dropped dx bit for bit; dgamma / dbeta up to the order of their float atomics."""
    H, rows = 768, (5 * 86, 5 * 41 + 3)
    xs = tuple(_rand((r, H), dtype, 81 + i, 1.0) for i, r in enumerate(rows))
    dys = tuple(_rand((r, H), dtype, 83 + i, 1.0) for i, r in enumerate(rows))
    gs = tuple(_rand((H,), torch.float32, 85 + i, 1.0) for i in range(2))
    bs = tuple(_rand((H,), torch.float32, 87 + i, 1.0) for i in range(2))
    singles = [ops.ln_fwd(xs[i], gs[i], bs[i], 1e-12) for i in range(2)]
    duals = ops.ln_fwd2(xs, gs, bs, 1e-12)
    for i in range(2):
        for a, b in zip(singles[i], duals[i]):
            assert torch.equal(a, b)
    means, rstds = tuple(s[1] for s in singles), tuple(s[2] for s in singles)
    drop1 = [ops.ln_bwd(dys[i], xs[i], gs[i], means[i], rstds[i], drop=(p_drop, 11 + i)) for i in range(2)]
    drop2 = ops._ln_bwd_to2(dys, xs, gs, bs, means, rstds, (True, True), drop=(p_drop, (11, 12)))
    for i in range(2):
        assert torch.equal(drop1[i][0], drop2[i][0]) and torch.equal(drop1[i][3], drop2[i][3])
        for a, b in ((drop1[i][1], drop2[i][1]), (drop1[i][2], drop2[i][2])):
            assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item())
    none = ops._ln_bwd_to2(dys, xs, gs, bs, means, rstds, (False, True))
    assert none[0][1] is None and none[1][1] is not None and torch.equal(none[0][0], ops.ln_bwd(dys[0], xs[0], gs[0], means[0], rstds[0])[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_bias_residual_layernorm_entry_points(ops, dtype, eps):
    """vlni_bias_residual_layernorm_fwd/bwd through the C ABI (raw pointers) against LayerNorm(dropout_0(x + bias) + residual) in torch:
    the BertSelfOutput / BertOutput tail, VLN-HAMT/finetune_src/models/vilmodel_cmt.py:144-148,186-190."""
    from vln_imagine_amd import _lib
    torch.manual_seed(5)
    rows, H = 333, 768
    x0, r0 = (torch.randn(rows, H, device="cuda") * 0.7).to(dtype), (torch.randn(rows, H, device="cuda") * 0.7).to(dtype)
    bias, gamma, beta = (torch.randn(H, device="cuda") * s + o for s, o in ((0.1, 0.0), (0.2, 1.0), (0.1, 0.0)))
    dy = (torch.randn(rows, H, device="cuda") * 0.5).to(dtype)
    dtc = ops._DT[dtype]
    st = torch.cuda.current_stream().cuda_stream
    y, xs = torch.empty_like(x0), torch.empty_like(x0)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    _lib.call("vlni_bias_residual_layernorm_fwd", dtc, x0.data_ptr(), H, bias.data_ptr(), r0.data_ptr(), H, gamma.data_ptr(), beta.data_ptr(), eps,
              y.data_ptr(), H, xs.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), rows, H, st)
    dx = torch.empty_like(x0)
    dg, db, dbias = (torch.zeros(H, device="cuda") for _ in range(3))
    _lib.call("vlni_bias_residual_layernorm_bwd", dtc, dy.data_ptr(), H, xs.data_ptr(), H, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
              dx.data_ptr(), H, dg.data_ptr(), db.data_ptr(), dbias.data_ptr(), rows, H, st)
    xr, rr = x0.double().requires_grad_(), r0.double().requires_grad_()
    br, gr, ber = bias.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr + br + rr, (H,), gr, ber, eps)
    ref.backward(dy.double())
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert (y.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (dx.double() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())
    assert torch.equal(xr.grad, rr.grad)
    wtol = 1e-4 if dtype == torch.float32 else 5e-2
    for got, want in ((dg, gr.grad), (db, ber.grad), (dbias, br.grad)):
        assert (got.double() - want).abs().max().item() <= wtol * max(1.0, want.abs().max().item())
    # without bias / residual / saved sum: plain LayerNorm
    _lib.call("vlni_bias_residual_layernorm_fwd", dtc, x0.data_ptr(), H, 0, 0, 0, gamma.data_ptr(), beta.data_ptr(), eps,
              y.data_ptr(), H, 0, 0, mean.data_ptr(), rstd.data_ptr(), rows, H, st)
    ref2 = torch.nn.functional.layer_norm(x0.double(), (H,), gamma.double(), beta.double(), eps)
    assert (y.double() - ref2).abs().max().item() <= tol * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("variant", [14, 15, 32])
def test_timed_gemm_kernels_against_float64_at_bench_shapes(ops, variant, dtype):
    """The kernels bench.py times - persistent 128 x 128 (14), 256 x 256 8-phase (15), 256 x 128 with loader waves (32) - held to float64 math
    DIRECTLY (not through bit-identity with variant 1) at the bench's own launches: one step's two streams (5504 + 2752 rows, a dual
    launch) and the episode-batched backward (49536 rows), N in {768, 2304, 3072}, K in {768, 3072}, every epilogue kind the step uses.
    Bound: TOL x max |reference| (16-bit output rounding + the 16-bit epilogue operands)."""
    t = TOL[dtype]
    rows = {14: ((5504, 2752),), 15: ((5504, 2752), (33024, 16512)), 32: ((5504, 2752), (33024, 16512))}[variant]
    for (M0, M1) in rows:
        for (N, K) in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)):
            a = (_rand((M0, K), dtype, 61, 0.5), _rand((M1, K), dtype, 62, 0.5))
            b = (_rand((N, K), dtype, 63, 0.05), _rand((N, K), dtype, 64, 0.05))
            bias = (_rand((N,), torch.float32, 65, 0.1), _rand((N,), torch.float32, 66, 0.1))
            res = (_rand((M0, N), dtype, 67, 0.5), _rand((M1, N), dtype, 68, 0.5))
            zsrc = (_rand((M0, N), dtype, 69, 1.0), _rand((M1, N), dtype, 70, 1.0))
            lin = [x.double() @ w.double().t() for x, w in zip(a, b)]
            kinds = {"bias": dict(bias=bias), "bias + residual": dict(bias=bias, residual=res),
                     "bias + GELU + stored pre-activation": dict(bias=bias, act=1, want_z=True),
                     "bias + GELU + stored GELU' (act 3)": dict(bias=bias, act=3, want_z=True),
                     "GELU' of a stored pre-activation (dgrad)": dict(dact_src=zsrc, dact=1), "stored derivative (dgrad, dact 3)": dict(dact_src=zsrc, dact=3),
                     "residual (dgrad)": dict(residual=res)}
            for kind, kw in kinds.items():
                want_z = kw.pop("want_z", False)
                z = tuple(torch.empty_like(r) for r in res) if want_z else (None, None)
                saved = (ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS)
                try:
                    ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = True, (variant,), 1 << 30, 1 << 30
                    ops._GEMM_BEST.clear()
                    outs = ops.gemm_nt2(a, b, preact=z, **kw)
                finally:
                    ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = saved
                    ops._GEMM_BEST.clear()
                for i in range(2):
                    pre = lin[i] + (kw["bias"][i].double() if "bias" in kw else 0.0)
                    ref = pre
                    stored = pre
                    if kw.get("act") in (1, 3):
                        ref = torch.nn.functional.gelu(pre)
                    if kw.get("act") == 3:
                        zz = pre.clone().requires_grad_(True)
                        stored = torch.autograd.grad(torch.nn.functional.gelu(zz).sum(), zz)[0]
                    if kw.get("dact") == 1:
                        zz = zsrc[i].double().requires_grad_(True)
                        ref = pre * torch.autograd.grad(torch.nn.functional.gelu(zz).sum(), zz)[0]
                    if kw.get("dact") == 3:
                        ref = pre * zsrc[i].double()
                    if "residual" in kw:
                        ref = ref + res[i].double()
                    _chk(outs[i], ref, t, (variant, M0, N, K, kind, i))
                    if want_z:
                        _chk(z[i], stored, t, (variant, M0, N, K, kind + " (z)", i))


@pytest.mark.parametrize("variant", [7, 8])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ring_weight_gradient_against_float64_at_bench_shapes(ops, dtype, variant):
    """gemm_tn_ring_kernel (variants 7 and 8 = its two wave rows one barrier apart; what the flush picks for the episode-long reductions) against float64: 49536 rows as the two
    segments of a shared cross-attention weight, N x K in {2304, 3072, 768} x 768 and 768 x 3072, partial slabs + the batched reduction."""
    import ctypes
    import numpy as np
    from vln_imagine_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    Ms = [33024, 16512]
    for (N, K, split) in ((2304, 768, 9), (3072, 768, 7), (768, 768, 28), (768, 3072, 7)):
        dd = [_rand((m, N), dtype, 71 + i, 0.1) for i, m in enumerate(Ms)]
        xx = [_rand((m, K), dtype, 81 + i, 0.5) for i, m in enumerate(Ms)]
        ref = sum(d.double().t() @ x.double() for d, x in zip(dd, xx))
        refb = sum(d.double().sum(0) for d in dd)
        n, nmt = len(Ms), sum((m + 63) // 64 for m in Ms)
        eff, per = ops._eff_split(nmt, split)
        part = torch.empty((eff * (N * K + N),), device="cuda")
        gw, gb = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        pa = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dd]); pb = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xx])
        pm = (ctypes.c_int * n)(*Ms)
        cpart = part.data_ptr() + 4 * eff * N * K
        _lib.call("vlni_gemm_tn_h16_grouped_part", ops._DT[dtype], n, pa, pb, pm, N, K, part.data_ptr(), N * K, N, K, cpart, split, variant, st)
        arr = np.zeros((2,), ops._PART_DT)
        arr[0] = (gw.data_ptr(), part.data_ptr(), N * K // 4, N * K // 4, eff, 0)
        arr[1] = (gb.data_ptr(), cpart, N // 4, N // 4, eff, -(-(N * K // 4) // 1024))
        tab = torch.from_numpy(arr.view(np.uint8)).cuda()
        _lib.call("vlni_reduce_parts", tab.data_ptr(), 2, int(arr[1]["blk0"]) + -(-(N // 4) // 1024), st)
        # float32 accumulation of 49.5 k products of 16-bit inputs: no 16-bit rounding of the output at all
        _chk(gw, ref, 2e-5, ("dW", N, K))
        _chk(gb, refb, 2e-5, ("db", N, K))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_resident_backward_tile_edges(ops, dtype, p_drop):
    """The resident attention backward (<= 96 queries and keys: LDS-DMA staging, one 32 x 32 score tile per wave, independent output tiles)
    at the edges of its tiling - exactly 32 / 64 / 96 rows, one row past a tile, a single query or key, 1 to 9 score tiles, blocks of 1 to 6
    waves - with attention-probability dropout (the library's own mask, exported by vlni_dropout) and an additive bias, against float64."""
    from vln_imagine_amd import _lib
    H, nh, B = 768, 12, 2
    t = TOL[dtype]
    for (Sq, Sk, use_bias) in ((96, 96, False), (32, 32, True), (33, 31, False), (64, 65, True), (1, 96, False), (96, 1, False), (17, 49, True),
                               (65, 33, False), (86, 43, False), (43, 86, True)):
        qkv_q, qkv_k = _rand((B * Sq, 3 * H), dtype, 101, 0.7), _rand((B * Sk, 3 * H), dtype, 102, 0.7)
        lens = torch.tensor([Sk, max(1, Sk - 3)])
        kmask = ((torch.arange(Sk)[None, :] >= lens[:, None]).float() * -10000.0).cuda()
        bias = _rand((B, Sq, Sk), torch.float32, 103, 0.5) if use_bias else None
        q, k, v = qkv_q[:, :H], qkv_k[:, H:2 * H], qkv_k[:, 2 * H:]
        seed = 4242
        out, lse = ops.attn_fwd(q, k, v, B, Sq, Sk, kmask, bias, drop=(p_drop, seed))
        qr, kr, vr = (x.double().clone().requires_grad_(True) for x in (q, k, v))
        br = bias.double().clone().requires_grad_(True) if use_bias else None
        qq, kk, vv = (x.view(B, -1, nh, 64).transpose(1, 2) for x in (qr, kr, vr))
        s = qq @ kk.transpose(-1, -2) / 8.0 + kmask.double()[:, None, None, :]
        if use_bias:
            s = s + br[:, None]
        pr = torch.softmax(s, -1)
        if p_drop > 0:
            m = torch.empty((B, nh, Sq, Sk), dtype=torch.float32, device="cuda")      # mask(seed, ((b nh + h) Sq + q) Sk + key)
            _lib.call("vlni_dropout", 0, 0, m.data_ptr(), m.numel(), p_drop, seed, torch.cuda.current_stream().cuda_stream)
            pr = pr * m.double()
        ref = (pr @ vv).transpose(1, 2).reshape(B * Sq, H)
        _chk(out, ref, t, ("fwd", Sq, Sk))
        dout = _rand((B * Sq, H), dtype, 104)
        ref.backward(dout.double())
        dq_buf, dk_buf = torch.zeros_like(qkv_q), torch.zeros_like(qkv_k)
        dbias = torch.zeros_like(bias) if use_bias else None
        ops.attn_bwd(q, k, v, out, dout, lse, dq_buf[:, :H], dk_buf[:, H:2 * H], dk_buf[:, 2 * H:], B, Sq, Sk, kmask, bias, dbias, drop=(p_drop, seed))
        # a single key: P = 1, so dS = P (dP - delta) is EXACTLY zero and dK with it; the kernels take delta = <dO, O> from the 16-bit O, so what they
        # return for dK is the rounding of O summed over the queries - noise against a zero reference, not a fraction of it (same in the chunked kernel)
        tk = t * 4 if Sk == 1 else t
        _chk(dq_buf[:, :H], qr.grad, tk, ("dq", Sq, Sk))
        _chk(dk_buf[:, H:2 * H], kr.grad, tk, ("dk", Sq, Sk))
        _chk(dk_buf[:, 2 * H:], vr.grad, t, ("dv", Sq, Sk))
        assert float(dq_buf[:, H:].abs().max()) == 0.0 and float(dk_buf[:, :H].abs().max()) == 0.0      # nothing outside the slices
        if use_bias:
            _chk(dbias, br.grad, t * 5, ("dbias", Sq, Sk))


@pytest.mark.parametrize("variant", [5, 11, 14, 15, 32])
def test_dual_launch_with_mixed_epilogue_operands_takes_the_generic_kernel(ops, variant):
    """The persistent / large-tile kernels are instantiated per epilogue KIND and the launcher picks one for BOTH problems of a dual launch
    (epi_kind): problems that differ in what they carry - a residual or a bias on one stream only - must fall to the generic instantiation
    and still be right; so must dropout without a residual (no kind of its own)."""
    dtype, t = torch.bfloat16, TOL[torch.bfloat16]
    M0, M1, N, K = 1300, 1100, 768, 768
    a = (_rand((M0, K), dtype, 81, 0.5), _rand((M1, K), dtype, 82, 0.5))
    b = (_rand((N, K), dtype, 83, 0.05), _rand((N, K), dtype, 84, 0.05))
    bias = (_rand((N,), torch.float32, 85, 0.1), _rand((N,), torch.float32, 86, 0.1))
    res = (_rand((M0, N), dtype, 87, 0.5), _rand((M1, N), dtype, 88, 0.5))
    lin = [x.double() @ w.double().t() for x, w in zip(a, b)]
    cases = {"residual on problem 0 only": (dict(bias=bias, residual=(res[0], None)), [lin[0] + bias[0].double() + res[0].double(), lin[1] + bias[1].double()]),
             "residual on problem 1 only": (dict(bias=bias, residual=(None, res[1])), [lin[0] + bias[0].double(), lin[1] + bias[1].double() + res[1].double()]),
             "bias on problem 1 only": (dict(bias=(None, bias[1]), residual=res), [lin[0] + res[0].double(), lin[1] + bias[1].double() + res[1].double()])}
    saved = (ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS)
    try:
        ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = True, (variant,), 1 << 30, 1 << 30
        for what, (kw, refs) in cases.items():
            ops._GEMM_BEST.clear()
            outs = ops.gemm_nt2(a, b, **kw)
            for i in range(2):
                _chk(outs[i], refs[i], t, (variant, what, i))
        # dropout without a residual: every kept element is the plain result / (1 - p), the rest exactly zero, about p of them
        ops._GEMM_BEST.clear()
        outs = ops.gemm_nt2(a, b, bias=bias, drop=(0.25, (7, 8)))
        for i in range(2):
            ref = (lin[i] + bias[i].double()) / 0.75
            o = outs[i].double()
            kept = o != 0
            assert 0.70 < kept.double().mean().item() < 0.80, (variant, i, kept.double().mean().item())
            assert ((o - ref).abs()[kept]).max().item() < t * max(1.0, ref.abs().max().item()), (variant, i)
    finally:
        ops.AUTOTUNE, ops.GEMM_VARIANTS, ops.P8_MIN_ROWS, ops.P8H_MIN_ROWS = saved
        ops._GEMM_BEST.clear()


def test_table_rows_out_of_range_are_skipped_and_counted(ops):
    """vlni_embed_combine_fwd / vlni_scatter_add_rows* take the table's row count; a navigation-type / step id outside the table
    adds nothing (no out-of-bounds read, no out-of-bounds scatter in the backward) and is counted in the registered device counter, from which
    ops.index_errors() raises the IndexError the reference's nn.Embedding lookup (vilmodel_cmt.py:535-544, :596-618) would have raised."""
    torch.manual_seed(4)
    rows, H = 64, 768
    a = torch.randn(rows, H).cuda()
    table = (torch.randn(3, H) * 0.2).cuda().requires_grad_(True)
    big = (torch.randn(40, H) * 0.2).cuda().requires_grad_(True)
    idx = torch.randint(0, 3, (rows,)).cuda()
    idx2 = torch.randint(0, 40, (rows,)).cuda()
    ops.watch_index_errors("cuda")
    assert ops.index_errors() == 0
    good = ops.embed_combine(a, torch.float32, table=(table, idx), p_drop=0.0, training=False)
    good2 = ops.embed_combine(a, torch.float32, table=(big, idx2), p_drop=0.0, training=False)
    bad, bad2 = idx.clone(), idx2.clone()
    bad[5], bad[9], bad2[7] = 3, -1, 40                                      # one past the end, negative, one past the end of the large table
    y = ops.embed_combine(a, torch.float32, table=(table, bad), p_drop=0.0, training=False)
    y2 = ops.embed_combine(a, torch.float32, table=(big, bad2), p_drop=0.0, training=False)
    keep = torch.ones(rows, dtype=torch.bool, device="cuda")
    keep[5] = keep[9] = False
    assert torch.equal(y[keep], good[keep]) and torch.equal(y[~keep], a[~keep])      # the out-of-range rows got no table row
    assert torch.equal(y2[7], a[7])
    (y.sum() + y2.sum()).backward()                                           # backward: small-table and generic scatter kernels, both bounded
    assert torch.isfinite(table.grad).all() and torch.isfinite(big.grad).all()
    cnt = torch.bincount(idx[keep], minlength=3).float()
    assert torch.allclose(table.grad.sum(1), cnt * H, rtol=1e-5)              # rows 5 and 9 scattered nowhere
    assert ops.index_errors(raise_=False) == 6                                # 3 in the forward launches + 3 in the scatters
    with pytest.raises(IndexError):
        ops.embed_combine(a, torch.float32, table=(table, bad), p_drop=0.0, training=False)
        ops.index_errors()
    assert ops.index_errors() == 0                                            # reset by the raising call


def test_block_entry_points_reject_other_head_sizes(ops):
    """the block-level C entry points score with 1 / sqrt(64); H != 64 nh is VLNI_EUNSUP (-3) there and the Python side keeps such a
    configuration on the launch-by-launch path."""
    import ctypes
    from vln_imagine_amd import _lib
    a = ops._BlkArgs()
    a.dtype, a.n, a.H, a.FF, a.nh = 1, 1, 768, 0, 8                          # 96-wide heads
    lib = _lib.load()
    assert lib.vlni_self_att_block_fwd(ctypes.addressof(a), 0) == -3
    assert lib.vlni_self_att_block_bwd(ctypes.addressof(a), 0) == -3
    x = torch.randn(2, 16, 768).cuda().bfloat16()
    assert ops._blk_self_att_fwd([(x, None, (0.0, 0.0), None)], None, 1e-12, nh=8) is None
