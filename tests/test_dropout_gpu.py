"""GPU: train-mode dropout inside the fused kernels. The mask is a pure function of (seed, element index), exported by
vlni_dropout, so a torch fp64 reference with the SAME masks must match forward and backward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mask(ops, shape, p, seed):
    from vln_imagine_amd import _lib
    m = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.call("vlni_dropout", 0, 0, m.data_ptr(), m.numel(), p, seed, torch.cuda.current_stream().cuda_stream)
    return m.double()


def _err(a, b):
    return (a.double() - b.double()).abs().max().item()


def test_mask_statistics_and_determinism():
    from vln_imagine_amd import ops
    m = _mask(ops, (1 << 20,), 0.1, 1234)
    keep = (m > 0).double().mean().item()
    assert abs(keep - 0.9) < 2e-3 and abs(m.max().item() - 1 / 0.9) < 1e-6
    assert torch.equal(m, _mask(ops, (1 << 20,), 0.1, 1234)) and not torch.equal(m, _mask(ops, (1 << 20,), 0.1, 1235))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_blocks_with_dropout_match_torch_with_same_masks(dtype):
    from vln_imagine_amd import ops
    torch.manual_seed(0)
    B, S, Sv, H, FF, nh = 2, 40, 21, 768, 3072, 12
    pa, ph, seed = 0.1, 0.2, 777
    mk = lambda *s, sc=0.04: (torch.randn(*s) * sc).cuda().requires_grad_(True)
    att = [mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1)]
    g = (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True)
    b = (0.1 * torch.randn(H)).cuda().requires_grad_(True)
    ffn = [mk(FF, H), mk(FF, sc=0.1), mk(H, FF, sc=0.02), mk(H, sc=0.1)]
    params = att + [g, b] + ffn
    x = torch.randn(B, S, H).cuda().to(dtype)
    ctxt = torch.randn(B, Sv, H).cuda().to(dtype)
    km = ((torch.rand(B, S) > 0.2).float())
    km[:, 0] = 1
    akm = ((1 - km) * -10000.0).cuda()
    akv = torch.zeros(B, Sv).cuda()

    def ref_att(xq, c, mask, pmask, hmask):
        wq, bq, wk, bk, wv, bv, wo, bo = [w.double() for w in att]
        Bq, Sq, _ = xq.shape
        Sk = c.shape[1]
        q = F.linear(xq, wq, bq).view(Bq, Sq, nh, 64).transpose(1, 2)
        k = F.linear(c, wk, bk).view(Bq, Sk, nh, 64).transpose(1, 2)
        v = F.linear(c, wv, bv).view(Bq, Sk, nh, 64).transpose(1, 2)
        pr = torch.softmax(q @ k.transpose(-1, -2) / 8 + mask.double()[:, None, None, :], -1) * pmask
        a = (pr @ v).transpose(1, 2).reshape(Bq, Sq, H)
        return F.layer_norm(F.linear(a, wo, bo) * hmask + xq, (H,), g.double(), b.double(), 1e-12)

    def ref_ffn(xx, hmask):
        w1, b1, w2, b2 = [w.double() for w in ffn]
        return F.layer_norm(F.linear(F.gelu(F.linear(xx, w1, b1)), w2, b2) * hmask + xx, (H,), g.double(), b.double(), 1e-12)

    tol = 1e-4 if dtype == torch.float32 else 6e-2
    gs = 1.0 if dtype == torch.float32 else 40.0

    def run(prod, ref, inputs):
        xp = [t.detach().clone().requires_grad_(True) for t in inputs]
        xr = [t.detach().double().requires_grad_(True) for t in inputs]
        for prm in params:
            prm.grad = None
        op, orr = prod(*xp), ref(*xr)
        assert _err(op, orr) < tol, ("fwd", _err(op, orr))
        w = torch.randn_like(orr)
        (op.double() * w).sum().backward()
        gp = [prm.grad.clone() for prm in params if prm.grad is not None]
        gx = [t.grad.clone() for t in xp]
        for prm in params:
            prm.grad = None
        (orr * w).sum().backward()
        gr = [prm.grad.clone() for prm in params if prm.grad is not None]
        for a_, r_ in zip(gx, [t.grad for t in xr]):
            assert _err(a_, r_) < tol * gs * max(1.0, r_.abs().max().item()), ("dx", _err(a_, r_))
        assert len(gp) == len(gr)
        for a_, r_ in zip(gp, gr):
            assert _err(a_, r_) < tol * gs * max(1.0, r_.abs().max().item()), ("dparam", _err(a_, r_))

    P = tuple(att) + (g, b)
    PF = tuple(ffn) + (g, b)
    run(lambda t: ops.self_att_block(t, akm, P, drop=(pa, ph, seed)),
        lambda t: ref_att(t, t, akm, _mask(ops, (B, nh, S, S), pa, seed), _mask(ops, (B, S, H), ph, seed + 1)), [x])
    run(lambda t: ops.ffn_block(t, PF, drop=(0.0, ph, seed)), lambda t: ref_ffn(t, _mask(ops, (B, S, H), ph, seed)), [x])
    run(lambda t, c: ops.xatt_block(t, c, akv, P, drop=(pa, ph, seed)),
        lambda t, c: ref_att(t, c, akv, _mask(ops, (B, nh, S, Sv), pa, seed), _mask(ops, (B, S, H), ph, seed + 1)), [x, ctxt])


def test_models_train_mode_runs_and_is_reproducible():
    from tests.golden.variants import duet_variant_setup, hamt_variant_setup
    from tests.test_duet_gpu import build_product as build_duet
    from tests.test_hamt_gpu import build_product as build_hamt
    from vln_imagine_amd import ops
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode as run_duet
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode as run_hamt
    for build, setup, name, run, ET in ((build_hamt, hamt_variant_setup, "c1_language", run_hamt, EpisodeTensors),
                                        (build_duet, duet_variant_setup, "c1_shipped", run_duet, DuetEpisodeTensors)):
        cfg, ep = setup(name)
        et = ET(ep, "cuda")
        losses = []
        for rep in range(2):
            m = build(cfg).train()
            torch.manual_seed(5); ops.reseed(99)
            out = run(m, et)
            out["loss"].backward()
            assert torch.isfinite(out["loss"])
            gn = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None)
            assert gn > 0 and gn == gn
            losses.append(out["loss"].item())
        assert losses[0] == losses[1], losses                 # same seeds -> same masks -> same loss
        m.eval()
        assert abs(run(m, et)["loss"].item() - losses[0]) > 1e-4      # dropout really was active in train mode


@pytest.mark.parametrize("train", [False, True])
def test_dual_stream_blocks_equal_two_single_blocks(train):
    """The dual-problem launches (language + vision stream in one GEMM launch) reproduce the single-stream blocks
    bit for bit in forward, and to rounding in backward (with the same dropout seeds in train mode)."""
    from vln_imagine_amd import ops
    torch.manual_seed(1)
    B, S0, S1, H, FF = 3, 45, 20, 768, 3072
    mk = lambda *s, sc=0.04: (torch.randn(*s) * sc).cuda().requires_grad_(True)
    def att():
        return [mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1), mk(H, H), mk(H, sc=0.1),
                (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True), (0.1 * torch.randn(H)).cuda().requires_grad_(True)]
    def ffn():
        return [mk(FF, H), mk(FF, sc=0.1), mk(H, FF, sc=0.02), mk(H, sc=0.1),
                (1 + 0.1 * torch.randn(H)).cuda().requires_grad_(True), (0.1 * torch.randn(H)).cuda().requires_grad_(True)]
    A0, A1, F0, F1 = att(), att(), ffn(), ffn()
    params = A0 + A1 + F0 + F1
    x0, x1 = torch.randn(B, S0, H).cuda(), torch.randn(B, S1, H).cuda()
    m0, m1 = torch.zeros(B, S0).cuda(), torch.zeros(B, S1).cuda()
    m0[:, -5:] = -10000.0
    d0 = (0.1, 0.1, 4000) if train else ops.NO_DROP
    d1 = (0.1, 0.1, 5000) if train else ops.NO_DROP
    def single(a, b):
        y0 = ops.ffn_block(ops.self_att_block(a, m0, tuple(A0), drop=d0), tuple(F0), drop=(0.0, d0[1], d0[2] + 7))
        y1 = ops.ffn_block(ops.self_att_block(b, m1, tuple(A1), drop=d1), tuple(F1), drop=(0.0, d1[1], d1[2] + 7))
        return y0, y1
    def dual(a, b):
        y0, y1 = ops.dual_self_att_block(a, b, m0, m1, tuple(A0), tuple(A1), drop0=d0, drop1=d1)
        return ops.dual_ffn_block(y0, y1, tuple(F0), tuple(F1), drop0=(0.0, d0[1], d0[2] + 7), drop1=(0.0, d1[1], d1[2] + 7))
    res = []
    for fn in (single, dual):
        a, b = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
        for prm in params:
            prm.grad = None
        y0, y1 = fn(a, b)
        (y0.sum() * 0.7 + (y1 * y1).sum()).backward()
        res.append((y0.detach(), y1.detach(), a.grad.clone(), b.grad.clone(), [prm.grad.clone() for prm in params]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for i in (2, 3):
        assert _err(res[0][i], res[1][i]) < 1e-5
    for g0, g1 in zip(res[0][4], res[1][4]):
        assert _err(g0, g1) < 1e-4 * max(1.0, g0.abs().max().item())


def test_seed_base_shifts_every_seed():
    """vlni_set_dropout_seed_base: mask(seed, base) == mask(seed + base, no base), for the mask kernel and a fused block."""
    from vln_imagine_amd import ops
    try:
        ops.set_seed_base(None)                   # a GraphedStep of an earlier test may still be registered
        ref = _mask(ops, (4096,), 0.3, 1000 + 77)
        base = torch.full((1,), 77, dtype=torch.int32, device="cuda")
        ops.set_seed_base(base)
        assert torch.equal(_mask(ops, (4096,), 0.3, 1000), ref)
        base.add_(1)
        assert not torch.equal(_mask(ops, (4096,), 0.3, 1000), ref)
        # a fused GEMM epilogue and an attention launch follow the base too
        a = torch.randn(256, 768, device="cuda").bfloat16(); w = (torch.randn(768, 768, device="cuda") * 0.05).bfloat16()
        q = torch.randn(2 * 40, 768, device="cuda").bfloat16()
        km = torch.zeros(2, 40, device="cuda")
        y_b = ops.gemm_nt(a, w, drop=(0.2, 500)).clone()
        o_b, _ = ops.attn_fwd(q, q, q, 2, 40, 40, km, drop=(0.1, 900))
        o_b = o_b.clone()
        ops.set_seed_base(None)
        y_s = ops.gemm_nt(a, w, drop=(0.2, 500 + 78))
        o_s, _ = ops.attn_fwd(q, q, q, 2, 40, 40, km, drop=(0.1, 900 + 78))
        assert torch.equal(y_b, y_s) and torch.equal(o_b, o_s)
    finally:
        ops.set_seed_base(None)


def test_graph_replay_draws_fresh_masks(monkeypatch):
    """A captured training step with dropout p > 0: every replay uses new masks (device seed base advanced in-graph) and equals
    the eager step run with the same base value and the same host seed counter. torch's own dropouts (small tensors: embedding
    outputs, heads) are switched off here: their Philox offsets are torch's business, the fused kernels' seeds are ours."""
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    from tests.golden.variants import hamt_variant_setup
    from tests.test_hamt_gpu import build_product
    from vln_imagine_amd import ops
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = hamt_variant_setup("c1_T3_dense")
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.1
    et = EpisodeTensors(ep, "cuda")
    try:
        m = build_product(cfg).train()
        tr = FlatTrainer(m, lr=1e-3)

        def fwd_bwd(model=None):
            ops.reseed(4242)                      # the host counter only matters while recording
            loss = run_episode(model or m, et, criterion=ops.cross_entropy_sum, keep=False)["loss"]
            loss.backward()
            return loss

        step = tr.capture(fwd_bwd, warmup=1)
        p0, m0, v0, st0, gs0 = tr.flat_p.clone(), tr.m.clone(), tr.v.clone(), tr.state.clone(), tr.gstate.clone()
        l1 = float(step().detach())
        base1 = int(step.seed_base.item())
        p1 = tr.flat_p.clone()
        l2 = float(step().detach())
        assert int(step.seed_base.item()) == base1 + 7919 and abs(l1 - l2) > 1e-6      # new masks, new loss
        # eager twin: same start, same base value, same host seeds -> same first step
        m2 = build_product(cfg).train()
        tr2 = FlatTrainer(m2, lr=1e-3)
        tr2.flat_p.copy_(p0); tr2.m.copy_(m0); tr2.v.copy_(v0); tr2.state.copy_(st0); tr2.gstate.copy_(gs0)
        ops.SHADOWS.invalidate()
        step.seed_base.fill_(base1)
        tr2.zero_grad()
        le = fwd_bwd(m2)
        tr2.step()
        assert abs(float(le.detach()) - l1) < 2e-4 * max(1.0, abs(l1)), (float(le.detach()), l1)
        d = (tr2.flat_p - p1).abs()
        assert d.max().item() < 2.5e-3 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())
    finally:
        ops.set_seed_base(None)
        ops._WQ.clear()
