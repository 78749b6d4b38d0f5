"""GPU parity gate for DUET: the HIP GlocalTextPathNavCMT (fp32 compute) against the reference's golden vectors;
tolerance 1e-4 on logits and losses. bf16 path reported against fp32 (loose)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.variants import DUET_VARIANTS, duet_variant_setup
from tests.test_hamt_gpu import _close
from vln_imagine_amd import ops, synth
from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode
from vln_imagine_amd.duet.spec import param_shapes

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_product(cfg, dtype=torch.float32):
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
    m = GlocalTextPathNavCMT(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
    return m.cuda().eval().set_compute_dtype(dtype)


# The episode drivers, each against the reference's fixtures directly (see tests/test_hamt_gpu.py): stepwise = GMapNavAgent.rollout's call
# pattern (agent.py:409-500 + one backward), taped = step-by-step forward on maps padded to the episode's largest + one episode-batched
# backward (what bench.py --model duet times), time_batched = forward batched over time too, graph = the taped step captured and REPLAYED.
# dropin = the same calls through the reference-facing VLNBert wrapper (models/model.py: what an unchanged GMapNavAgent.rollout talks to)
DRIVERS = ("stepwise", "taped", "time_batched", "graph", "dropin")


def run_driver(driver, model, et):
    from vln_imagine_amd.duet.episode import run_episode_taped, run_episode_time_batched
    kw = dict(criterion=ops.cross_entropy_sum)
    if driver == "stepwise":
        out = run_episode(model, et, **kw)
        out["loss"].backward()
        return out, None
    if driver == "dropin":
        from vln_imagine_amd import dropin
        out = run_episode(dropin.wrap_duet(model, feat_dropout=0.0), et, **kw)
        out["loss"].backward()
        return out, None
    if driver == "time_batched":
        out = run_episode_time_batched(model, et, **kw)
        out["loss"].backward()
        return out, None
    if driver == "taped":
        out = run_episode_taped(model, et, **kw)
        out["loss"].backward()
        return out, None
    from vln_imagine_amd.train import FlatTrainer
    tr = FlatTrainer(model, lr=0.0, weight_decay=0.0)
    tape, stash = ops.EpisodeTape(et.T), {}

    def fwd_bwd():
        o = run_episode_taped(model, et, tape=tape, **kw)
        o["loss"].backward()
        stash["out"] = o
        return o["loss"]

    step = tr.capture(fwd_bwd, warmup=1)
    step()                                            # a REPLAY: its outputs and gradients are what is checked
    torch.cuda.synchronize()
    return stash["out"], tr


@pytest.mark.parametrize("driver", DRIVERS)
@pytest.mark.parametrize("name", list(DUET_VARIANTS))
def test_product_fp32_matches_reference_golden(name, driver, golden_dir):
    g = np.load(os.path.join(golden_dir, f"duet_{name}.npz"))
    cfg, ep = duet_variant_setup(name)
    model = build_product(cfg)
    tr = None
    try:
        out, tr = run_driver(driver, model, DuetEpisodeTensors(ep, "cuda"))
        c = lambda t: t.detach().float().cpu().numpy()
        _close(out["loss"].item(), g["loss"], TOL, "loss")
        _close(out["aux"].item(), g["aux"], TOL, "aux")
        _close(c(out["imagine_embeds"]), g["imagine_embeds"], TOL, "imagine_embeds")
        if "og_loss" in g:                                            # REVERIE: object grounding head
            _close(out["og_loss"].item(), g["og_loss"], TOL, "og_loss")
            for t in range(ep.T):
                _close(c(out["obj"][t]), g[f"obj{t}"], TOL, f"obj{t}")
        for t in range(ep.T):
            for nm in ("fused", "global", "local"):
                a, ref = c(out[nm][t]), g[f"{nm}{t}"]
                if a.shape[1] > ref.shape[1]:                     # padded drivers: columns beyond the step's own map / panorama are masked
                    assert np.isneginf(a[:, ref.shape[1]:]).all(), (nm, t)
                    a = a[:, :ref.shape[1]]
                _close(a, ref, TOL, f"{nm}{t}")
            for nm in ("pano", "gmap", "vp"):
                if nm in out:
                    a = c(out[nm][t])
                    want = g[f"{nm}{t}.shape"].tolist()
                    if list(a.shape) != want:                     # panorama padded to the episode's widest: compare the step's own columns
                        a = a[:, :want[1]]
                    _close(synth.probe(a)["samples"], g[f"{nm}{t}.samples"], TOL, f"{nm}{t}")
        if driver in ("taped", "graph"):
            for t in range(ep.T):
                a, ref = c(out["step_logits"][t]), g[f"fused{t}"]
                _close(a[:, :ref.shape[1]], ref, TOL, f"step_logits{t}")
        params = dict(model.named_parameters())
        for i, n in enumerate(g["grad_names"].tolist()):
            gr, ref_norm = params[n].grad, g["grad_norms"][i]
            if ref_norm < 0:
                assert gr is None or float(gr.abs().max()) == 0.0, n
                continue
            assert gr is not None, n
            nrm = float(gr.double().norm())
            assert abs(nrm - ref_norm) <= max(2e-4 * ref_norm, 2e-5), (n, nrm, ref_norm)   # 2e-5 abs: scalar grads that sum thousands of cancelling terms
            head = gr.reshape(-1)[:8].cpu().numpy()
            _close(head, g["grad_heads"][i][:head.size], 2e-4, f"grad {n}", rel=True)
    finally:
        if tr is not None:
            tr.close()
        ops.set_seed_base(None)
        ops._WQ.clear()


def test_product_bf16_tracks_fp32():
    cfg, ep = duet_variant_setup("c1_shipped")
    et = DuetEpisodeTensors(ep, "cuda")
    o32 = run_episode(build_product(cfg), et)
    o16 = run_episode(build_product(cfg, torch.bfloat16), et)
    assert abs(o16["loss"].item() - o32["loss"].item()) < 3e-2
    for t in range(ep.T):
        a, b = o16["fused"][t].float(), o32["fused"][t].float()
        fin = torch.isfinite(b)
        assert (torch.isfinite(a) == fin).all()
        assert (a[fin] - b[fin]).abs().max().item() < 0.15


def test_text_kv_cache_is_transparent():
    """SURVEY.md 8f rank 1 (DUET half): projecting the step-invariant text K/V once per episode must not change anything."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.duet.models import vilmodel
    cfg, ep = duet_variant_setup("c1_shipped")
    et = DuetEpisodeTensors(ep, "cuda")
    res = {}
    for on in (False, True):
        vilmodel.CACHE_TEXT_KV = on
        try:
            model = build_product(cfg)
            outs = []
            for rep in range(2):                      # second pass: the entry of pass 1 was dropped by its backward
                model.zero_grad(set_to_none=True)
                out = run_episode(model, et, criterion=ops.cross_entropy_sum)
                out["loss"].backward()
                outs.append(out)
            assert model._kv_cache is None
            assert abs(outs[0]["loss"].item() - outs[1]["loss"].item()) < 1e-6
            res[on] = (outs[1], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            vilmodel.CACHE_TEXT_KV = True
    (o0, g0), (o1, g1) = res[False], res[True]
    assert abs(o0["loss"].item() - o1["loss"].item()) < 1e-6
    for t in range(ep.T):
        a, b = o0["fused"][t], o1["fused"][t]
        fin = torch.isfinite(a)
        assert (a[fin] - b[fin]).abs().max().item() < 1e-5
    assert g0.keys() == g1.keys()
    for n in g0:
        d = (g0[n] - g1[n]).abs().max().item()
        assert d <= 1e-5 + 1e-4 * g0[n].abs().max().item(), (n, d)


def test_text_kv_cache_sees_parameter_updates():
    from vln_imagine_amd.duet.models import vilmodel
    cfg, ep = duet_variant_setup("c1_shipped")
    et = DuetEpisodeTensors(ep, "cuda")
    model = build_product(cfg)
    with torch.no_grad():
        a = run_episode(model, et)["fused"][-1].clone()
        w = model.global_encoder.encoder.x_layers[0].visual_attention.att.key.weight
        w.mul_(1.5)
        b = run_episode(model, et)["fused"][-1]
    fin = torch.isfinite(a)
    assert (a[fin] - b[fin]).abs().max().item() > 1e-4


def test_logit_fusion_kernel_matches_reference_loop():
    """vlni_duet_fuse_fwd/bwd against a literal restatement of VLN-DUET/map_nav_src/models/vilmodel.py:1198-1217 (autograd)."""
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
    from vln_imagine_amd.duet.config import DuetConfig
    torch.manual_seed(0)
    B, G, V = 3, 7, 6
    gl0, ll0 = torch.randn(B, G), torch.randn(B, V)
    vpids = [[None, "a", "b", "c", "d", "e", "f"], [None, "a", "b", "c", "d"], [None, "x", "y", "z", "w", "u", "t"]]
    vis = torch.tensor([[0, 1, 1, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    cands = [[None, "b", "d", "e"], [None, "c"], [None, "x", "t", "q"]]
    w = torch.randn(B, G)

    def masked(gl, ll):
        gl = gl.masked_fill(vis.to(gl.device), -float("inf"))
        keep = torch.ones(B, V, dtype=torch.bool, device=ll.device)
        keep[:, 4:] = False
        return gl, ll.masked_fill(~keep, -float("inf"))

    gl_r, ll_r = gl0.clone().requires_grad_(), ll0.clone().requires_grad_()
    gl, ll = masked(gl_r, ll_r)
    rows = []
    for i in range(B):
        visited = {vp for vp, m in zip(vpids[i], vis[i].tolist()) if m}
        tmp, bw = {}, 0
        for j, c in enumerate(cands[i]):
            if j > 0:
                if c in visited:
                    bw = bw + ll[i, j]
                else:
                    tmp[c] = ll[i, j]
        row = [gl[i, 0] + ll[i, 0]]
        for j in range(1, G):
            vp = vpids[i][j] if j < len(vpids[i]) else None
            add = 0
            if j < len(vpids[i]) and vp not in visited:
                add = tmp[vp] if vp in tmp else bw
            row.append(gl[i, j] + add)
        rows.append(torch.stack(row))
    ref = torch.stack(rows)
    fin = torch.isfinite(ref)
    (ref[fin] * w[fin]).sum().backward()

    model = GlocalTextPathNavCMT(DuetConfig(num_l_layers=1, num_pano_layers=1, num_x_layers=1))
    gl_g, ll_g = gl0.clone().cuda().requires_grad_(), ll0.clone().cuda().requires_grad_()
    out = model._fuse(*masked(gl_g, ll_g), vpids, vis.cuda(), cands)
    assert (torch.isfinite(out).cpu() == fin).all() and torch.allclose(out.cpu()[fin], ref[fin], atol=1e-6)
    (out[fin.cuda()] * w.cuda()[fin.cuda()]).sum().backward()
    assert torch.allclose(gl_g.grad.cpu(), gl_r.grad, atol=1e-6) and torch.allclose(ll_g.grad.cpu(), ll_r.grad, atol=1e-6)


@pytest.mark.parametrize("with_f", [True, False])
def test_fused_logit_tail_matches_unfused_composition(with_f):
    """ops.duet_heads (one launch) == sigmoid, scalings, masked_fills and ops.duet_fuse composed in torch (the round-2 path, itself pinned
    against the literal reference loop above), all three outputs and the gradients of the three inputs; mixed use of the outputs in the
    loss (global only / local + fused) exercises the null-gradient arguments of the backward kernel."""
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
    torch.manual_seed(1)
    B, G, V = 5, 9, 7
    vpids = [[None] + [f"n{b}_{j}" for j in range(1, G)] for b in range(B)]
    cands = [[None] + [f"n{b}_{j}" for j in (2, 3, 5)] + ["zz", "n%d_7" % b, "qq"] for b in range(B)]
    vis = torch.zeros(B, G, dtype=torch.bool)
    vis[:, 2] = True; vis[1, 5] = True; vis[3, 7] = True
    gmask = torch.ones(B, G, dtype=torch.bool); gmask[2, 6:] = False
    nav = torch.ones(B, V, dtype=torch.bool); nav[:, 5] = False; nav[4, 3] = False
    src, bw = GlocalTextPathNavCMT.fuse_plan(vpids, vis.tolist(), cands, G, V)
    src, bw = torch.tensor(src, dtype=torch.int32).cuda(), torch.tensor(bw, dtype=torch.uint8).cuda()
    g0, l0, f0 = torch.randn(B, G), torch.randn(B, V), torch.randn(B)
    wts = [torch.randn(B, G).cuda(), torch.randn(B, V).cuda(), torch.randn(B, G).cuda()]
    visc, gmc, navc = vis.cuda(), gmask.cuda(), nav.cuda()

    def unfused(g, l, f):
        w = torch.sigmoid(f)[:, None] if f is not None else 0.5
        gl = (g * w).masked_fill(visc | ~gmc, -float("inf"))
        ll = (l * (1 - w)).masked_fill(~navc, -float("inf"))
        return gl, ll, ops.duet_fuse(gl, ll, src, bw)

    def fused(g, l, f):
        return ops.duet_heads(g, l, f, visc, gmc, navc, src, bw)

    for use in ((0, 1, 2), (0,), (1, 2), (2,)):
        grads = []
        for fn in (unfused, fused):
            g, l = g0.clone().cuda().requires_grad_(), l0.clone().cuda().requires_grad_()
            f = f0.clone().cuda().requires_grad_() if with_f else None
            outs = fn(g, l, f)
            loss = 0
            for i in use:
                fin = torch.isfinite(outs[i])
                loss = loss + (outs[i][fin] * wts[i][fin]).sum()
            loss.backward()
            z = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)          # an input no used output depends on
            grads.append(([o.detach() for o in outs], z(g), z(l), z(f) if with_f else None))
        (oa, ga, la, fa), (ob, gb, lb, fb) = grads
        for a, b in zip(oa, ob):
            assert torch.equal(torch.isfinite(a), torch.isfinite(b))
            fin = torch.isfinite(a)
            assert torch.allclose(a[fin], b[fin], atol=1e-6), use
        assert torch.allclose(ga, gb, atol=1e-6) and torch.allclose(la, lb, atol=1e-6), use
        if with_f:
            assert torch.allclose(fa, fb, atol=1e-5), use


def test_masked_sequence_mean():
    """ops.seq_mean with lens == the agent's masked panorama mean (r2r/agent.py:159-161), forward and backward, fp32 and bf16."""
    torch.manual_seed(2)
    B, S, H = 6, 36, 768
    lens = torch.tensor([36, 30, 1, 17, 36, 29]).cuda()
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        x0 = torch.randn(B, S, H, device="cuda").to(dt)
        w = torch.randn(B, H, device="cuda").to(dt)
        xa, xb = x0.clone().requires_grad_(), x0.clone().requires_grad_()
        a = ops.seq_mean(xa, lens)
        m = torch.arange(S, device="cuda")[None, :] < lens[:, None]
        b = (xb.float() * m.unsqueeze(2)).sum(1) / lens[:, None]
        assert torch.allclose(a.float(), b, atol=tol)
        (a.float() * w.float()).sum().backward(); (b * w.float()).sum().backward()
        assert torch.allclose(xa.grad.float(), xb.grad.float(), atol=tol)
        assert (xa.grad[2, 1:] == 0).all()
    xa = x0.float().clone().requires_grad_()
    assert torch.allclose(ops.seq_mean(xa), xa.mean(1), atol=1e-5)               # unmasked form unchanged


def test_bf16_trainer_step_with_odd_sized_parameters():
    """DUET has 1-element parameters (sprel_linear): the bf16 mirror of the arena must still hand out 16-byte aligned views."""
    from vln_imagine_amd import ops
    from vln_imagine_amd.train import FlatTrainer
    cfg, ep = duet_variant_setup("c1_shipped")
    et = DuetEpisodeTensors(ep, "cuda")
    try:
        m = build_product(cfg, torch.bfloat16)
        tr = FlatTrainer(m, lr=1e-4)
        losses = []
        for _ in range(2):
            tr.zero_grad()
            loss = run_episode(m, et, criterion=ops.cross_entropy_sum, keep=False)["loss"]
            loss.backward()
            tr.step()
            losses.append(float(loss.detach()))
        assert all(np.isfinite(losses)) and all(p.data_ptr() % 32 == 0 for p in tr.params)
        w = m.global_encoder.encoder.x_layers[0].visn_self_att.self.query.weight
        mirror = ops._w((w,), torch.bfloat16)
        assert mirror.data_ptr() % 16 == 0 and torch.equal(mirror, w.detach().bfloat16())     # AdamW keeps the mirror current
        # transposed (dgrad) shadows: all rebuilt by ONE batched launch after an optimizer step
        assert ops.SHADOWS._tr and ops.SHADOWS._tr_table is not None
        for lyr in m.global_encoder.encoder.x_layers:
            for w2 in (lyr.visn_inter.dense.weight, lyr.visn_output.dense.weight):
                wt = ops._w((w2,), torch.bfloat16, True)                        # the maintained W^T copy (a WT handle with VLNI_NN_DGRAD=1)
                wt = wt.resolve(64) if isinstance(wt, ops.WT) else wt
                assert torch.equal(wt, w2.detach().bfloat16().t())
    finally:
        ops._WQ.clear()
        ops.SHADOWS.set_arena(None, None)
