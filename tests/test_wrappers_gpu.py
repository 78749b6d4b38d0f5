"""GPU: the agent-facing wrappers: VLNBertCMT / VLNBert mode dispatch, feature dropout, history
masks from lengths, state vector, Critic, checkpoint key remapping (VLN-HAMT/finetune_src/models/model_HAMT.py:13-96,289-300,
vlnbert_init.py:4-83; VLN-DUET/map_nav_src/models/model.py:12-62)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _hamt_args(**kw):
    d = dict(image_feat_size=768, angle_feat_size=4, num_l_layers=1, num_x_layers=1, hist_enc_pano=True, hist_pano_num_layers=1,
             num_h_layers=0, fix_lang_embedding=False, fix_hist_embedding=False, fix_obs_embedding=False, no_lang_ca=False,
             act_pred_token="ob_txt", imagine_enc_pano=True, bypass_imag_encoder=True, use_cosine_aux_loss=True, aux_loss_type="cosine",
             concat_imagine_with="language", feat_dropout=0.4, dropout=0.5, bert_ckpt_file=None, dataset="r2r", tokenizer="bert")
    d.update(kw)
    return argparse.Namespace(**d)


def _fill(model):
    from vln_imagine_amd import synth
    sd = model.state_dict()
    model.load_state_dict({k: (torch.from_numpy(synth.init_param(k, v.shape)) if v.dtype.is_floating_point else v) for k, v in sd.items()})


def _hamt_models():
    # the HAMT and DUET packages both expose a top-level `models` in the reference layout; here they are imported by full path
    from vln_imagine_amd.hamt.models.model_HAMT import Critic, VLNBertCMT
    torch.manual_seed(0)
    w = VLNBertCMT(_hamt_args())
    _fill(w.vln_bert)
    return w.cuda(), Critic(_hamt_args()).cuda()


def test_hamt_wrapper_dispatch_masks_states_and_dropout():
    w, critic = _hamt_models()
    w.eval()
    m = w.vln_bert
    B, L, V, I = 3, 20, 9, 2
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    txt_ids = torch.randint(1, 1000, (B, L), generator=g).cuda()
    txt_masks = (torch.arange(L)[None, :] < torch.tensor([20, 12, 17])[:, None]).cuda()
    txt = w("language", txt_ids=txt_ids, txt_masks=txt_masks)
    assert torch.equal(txt, m("language", txt_ids=txt_ids, txt_masks=txt_masks))
    imag = r(B, I, 768)
    assert torch.equal(w("imagine", imagine_pano_img_feats=imag), m("imagine", imagine_pano_img_feats=imag, imagine_masks=None))   # eval: dropout off
    hist = [w("history").expand(B, -1)]
    hist.append(w("history", hist_img_feats=r(B, 768), hist_ang_feats=r(B, 4), hist_pano_img_feats=r(B, 36, 768),
                  hist_pano_ang_feats=r(B, 36, 4), ob_step=0))
    hist_lens = [2, 1, 2]                                                   # sample 1 has ended: its second history token is masked
    ob_img, ob_ang = r(B, V, 768), r(B, V, 4)
    nav = torch.tensor([[1, 1, 2] + [0] * (V - 3)] * B).cuda()
    ob_masks = torch.ones(B, V, dtype=torch.bool).cuda()
    imasks = torch.ones(B, I, dtype=torch.bool).cuda()
    kw = dict(txt_embeds=txt, txt_masks=txt_masks, ob_img_feats=ob_img, ob_ang_feats=ob_ang, ob_nav_types=nav, ob_masks=ob_masks,
              imagine_embeds=imag, imagine_masks=imasks)
    out = w("visual", hist_embeds=hist, hist_lens=hist_lens, **kw)
    assert isinstance(out, tuple) and len(out) == 1                         # (logits,)
    hm = torch.tensor([[True, True], [True, False], [True, True]]).cuda()
    logits, txt_o, hist_o, ob_o = m("visual", hist_embeds=torch.stack(hist, 1), hist_masks=hm, **kw)
    assert torch.equal(out[0], logits)
    lg, states = w("visual", hist_embeds=hist, hist_lens=hist_lens, return_states=True, **kw)
    assert torch.equal(lg, logits) and torch.equal(states, txt_o[:, 0] * hist_o[:, 0])
    w.args.no_lang_ca = True
    assert torch.equal(w("visual", hist_embeds=hist, hist_lens=hist_lens, return_states=True, **kw)[1], hist_o[:, 0])
    w.args.no_lang_ca = False
    # visualisation variant of the tuple (model_HAMT.py:80-95): (logits, [states,] cross_attn_probs, self_attn_probs)
    viz = w("visual", hist_embeds=hist, hist_lens=hist_lens, return_cross_attention_probs=True, **kw)
    assert len(viz) == 3 and torch.equal(viz[0], logits) and len(viz[1]) == len(viz[2]) == 1
    assert viz[1][0][0].shape == (B, 12, L + I, 2 + V) and viz[2][0][1].shape == (B, 12, 2 + V, 2 + V)
    viz = w("visual", hist_embeds=hist, hist_lens=hist_lens, return_states=True, return_cross_attention_probs=True, **kw)
    assert len(viz) == 4 and torch.equal(viz[1], states)
    with pytest.raises(NotImplementedError):
        w("panorama")
    # train mode: the wrapper's feature dropout (p = feat_dropout) hits the caller's FEATURES, nothing else
    w.train()
    big = torch.ones(64, 4, 768).cuda()
    seen = []
    orig = m.forward
    m.forward = lambda mode, **k: seen.append(k["imagine_pano_img_feats"]) or k["imagine_pano_img_feats"]
    try:
        w("imagine", imagine_pano_img_feats=big)
    finally:
        m.forward = orig
    x = seen[0]
    zeros = float((x == 0).float().mean())
    assert abs(zeros - 0.4) < 0.02 and torch.allclose(x[x != 0], torch.tensor(1 / 0.6).cuda())
    # Critic = Linear ReLU Dropout Linear, squeezed
    critic.eval()
    s = r(5, 768)
    ref = critic.state2value(s).squeeze()
    assert torch.allclose(critic(s), ref, atol=1e-5) and critic(s).shape == (5,)


def test_hamt_checkpoint_key_remapping(tmp_path):
    """vlnbert_init.py:62-76: 'module.' prefixes are stripped and 'next_action*' keys move under 'bert.'."""
    from vln_imagine_amd.hamt.models.vlnbert_init import get_vlnbert_models
    m0 = get_vlnbert_models(_hamt_args())
    sd = {k: torch.full_like(v, 0.25) if v.dtype.is_floating_point else v for k, v in m0.state_dict().items()}
    ckpt = {}
    for i, (k, v) in enumerate(sd.items()):
        ckpt[("module." + k) if i % 2 == 0 else k] = v
    path = str(tmp_path / "ckpt.pt")
    torch.save(ckpt, path)
    m1 = get_vlnbert_models(_hamt_args(bert_ckpt_file=path))
    for k, v in m1.state_dict().items():
        if v.dtype.is_floating_point:
            assert float(v.min()) == 0.25 == float(v.max()), k


def test_duet_wrapper_dispatch_and_dropout():
    from vln_imagine_amd.duet.models.model import Critic, VLNBert
    args = argparse.Namespace(feat_dropout=0.4, dropout=0.5, bert_ckpt_file=None, dataset="r2r", tokenizer="bert", num_l_layers=1,
                              num_pano_layers=1, num_x_layers=1, enc_full_graph=True, graph_sprels=True, fusion="dynamic",
                              image_feat_size=768, angle_feat_size=4, obj_feat_size=0, imagine_enc_pano=True, bypass_imag_encoder=True,
                              use_cosine_aux_loss=False, fix_lang_embedding=False, fix_pano_embedding=False, fix_local_branch=False)
    w = VLNBert(args).cuda()
    seen = []
    w.vln_bert.forward = lambda mode, batch: seen.append((mode, batch)) or "ok"
    w.train()
    x = torch.ones(16, 36, 768).cuda()
    assert w("panorama", {"view_img_fts": x, "loc_fts": None}) == "ok"
    mode, batch = seen[-1]
    assert mode == "panorama" and batch["obj_img_fts"] is None and batch["never_given"] is None       # defaultdict(lambda: None)
    assert abs(float((batch["view_img_fts"] == 0).float().mean()) - 0.4) < 0.02 and batch["view_img_fts"] is not x
    w("navigation", {"txt_embeds": x})
    assert seen[-1][1]["txt_embeds"] is x                                      # no dropout outside 'panorama'
    with pytest.raises(NotImplementedError):
        w("visual", {})
    c = Critic(args).cuda().eval()
    s = torch.randn(7, 768).cuda()
    assert torch.allclose(c(s), c.state2value(s).squeeze(), atol=1e-5)


@pytest.mark.parametrize("family", ["hamt", "duet"])
def test_agent_backward_groups_weight_gradients_and_recasts_weights_in_one_launch(family):
    """What an unchanged reference agent gets from the wrappers' models (vln_imagine_amd/dropin.py restates its iteration: agent_cmt.py:809-832,
    agent_base.py:223-228): the weight gradients of every projection are computed over ALL steps in grouped launches by a callback at the end of
    loss.backward() (ops.GradSession) instead of one launch per step through autograd, and the 16-bit weight copies that torch.optim.AdamW's step
    leaves stale are re-cast in one launch (ShadowCache._refresh_plain). Same gradients, same losses over iterations (set_to_none on and off, two
    backward passes into one step), a fraction of the launches."""
    from tests.golden.variants import DUET_C1, HAMT_C1
    from vln_imagine_amd import dropin, ops, synth
    if family == "hamt":
        from tests.test_hamt_gpu import build_product
        from vln_imagine_amd.hamt.config import HamtConfig
        from vln_imagine_amd.hamt.episode import EpisodeTensors
        cfg = HamtConfig(**HAMT_C1)
        et = EpisodeTensors(synth.HamtEpisode(tag="agent", B=16, L=80, V=37, I=4, T=3, ragged=True), "cuda")
        wrap, loss_of = dropin.wrap_hamt, dropin.hamt_agent_loss
    else:
        from tests.test_duet_gpu import build_product
        from vln_imagine_amd.duet.config import DuetConfig
        from vln_imagine_amd.duet.episode import DuetEpisodeTensors
        cfg = DuetConfig(**DUET_C1)
        et = DuetEpisodeTensors(synth.DuetEpisode(tag="agent", B=16, L=80, V=36, I=4, T=3, ragged=True), "cuda")
        wrap, loss_of = dropin.wrap_duet, dropin.duet_agent_loss

    from vln_imagine_amd import graphed

    def program(on):
        # (per-call graphs off: they draw their dropout masks from seeds fixed at capture time, and this test compares the two gradient
        # protocols mask for mask; the graphs have their own tests below)
        was = ops.AUTO_DEFER, ops.BATCH_SHADOWS, graphed.ENABLED
        ops.AUTO_DEFER = ops.BATCH_SHADOWS = on
        graphed.ENABLED = False
        calls, real = {}, ops._lib.call

        def counting(name, *a):
            calls[name] = calls.get(name, 0) + 1
            return real(name, *a)
        try:
            ops.reseed(21)
            torch.manual_seed(3)
            m = build_product(cfg, torch.bfloat16).train()
            w = wrap(m, feat_dropout=0.0)
            tr = dropin.DropInTrainer(w, et, family, lr=1e-5)
            out = []
            for it in range(4):
                if it == 3:
                    ops._lib.call = counting
                tr.opt.zero_grad(set_to_none=it != 1)
                # iteration 3 (HAMT): a rollout without the alignment head - its parameters get no gradient and must read None, as through autograd
                loss, _ = loss_of(w, et, use_aux=False) if (it == 3 and family == "hamt") else loss_of(w, et)
                loss.backward()
                if it == 2:
                    loss_of(w, et)[0].backward()
                out.append((float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
                # the weights move the same way in both programs (an optimizer would feed each program's own summation-order noise back
                # through Adam's normalisation): the next iteration's loss shows whether every 16-bit copy followed
                with torch.no_grad():
                    for i, p in enumerate(tr.params):
                        p.mul_(1.0 + 1e-3 * ((i % 5) - 2))
            ops._lib.call = real
            tr.opt.zero_grad()
            loss_of(w, et)[0].backward()
            torch.nn.utils.clip_grad_norm_(tr.params, 40.0)
            tr.opt.step()                                  # torch.optim.AdamW on gradients that are views of the packed buffers
            assert all(torch.isfinite(p).all() for p in tr.params)
            return out, calls
        finally:
            ops._lib.call = real
            ops.AUTO_DEFER, ops.BATCH_SHADOWS, graphed.ENABLED = was

    (ref, c0), (got, c1) = program(False), program(True)
    for it, ((l0, g0), (l1, g1)) in enumerate(zip(ref, got)):
        assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (it, l0, l1)
        assert set(g0) == set(g1), (it, set(g0) ^ set(g1))
        if it == 3 and family == "hamt":
            assert not any(n.startswith("contrastive_alignment_model") for n in g1) and len(g1) > 100
        top = max(v.abs().max().item() for v in g0.values())
        for n in g0:
            d = (g0[n].float() - g1[n].float()).abs().max().item()
            assert d <= 1e-4 * top, (it, n, d, top)
    wg = lambda c: sum(v for k, v in c.items() if "gemm_tn" in k)
    recast = lambda c: sum(v for k, v in c.items() if k in ("vlni_cast", "vlni_transpose"))
    assert wg(c1) <= 0.6 * wg(c0), (wg(c0), wg(c1))          # T = 3 steps here (and the text encoder's projections run once per episode anyway)
    assert recast(c1) + 20 <= recast(c0) and c1.get("vlni_shadow_refresh", 0) in (1, 2), (recast(c0), recast(c1), c1.get("vlni_shadow_refresh"))


def test_gradient_session_survives_an_aborted_backward():
    """A backward pass that raises half way never reaches the engine callback that closes ops.GradSession: the next pass must find the stale
    session, drop what it queued, and produce the same gradients as plain autograd."""
    from tests.golden.variants import HAMT_C1
    from tests.test_hamt_gpu import build_product
    from vln_imagine_amd import dropin, ops, synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors
    cfg = HamtConfig(**HAMT_C1)
    et = EpisodeTensors(synth.HamtEpisode(tag="abort", B=8, L=80, V=37, I=4, T=2, ragged=True), "cuda")

    def grads(on, sabotage):
        was = ops.AUTO_DEFER
        ops.AUTO_DEFER = on
        real = ops.ln_bwd
        try:
            ops.reseed(5)
            m = build_product(cfg, torch.bfloat16)
            w = dropin.wrap_hamt(m, feat_dropout=0.0)
            if sabotage:
                calls = [0]

                def failing(*a, **k):
                    calls[0] += 1
                    if calls[0] == 3:
                        raise RuntimeError("injected")
                    return real(*a, **k)
                ops.ln_bwd = failing
                with pytest.raises(RuntimeError, match="injected"):
                    dropin.hamt_agent_loss(w, et)[0].backward()
                ops.ln_bwd = real
                for p in m.parameters():
                    p.grad = None
            dropin.hamt_agent_loss(w, et)[0].backward()
            return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.ln_bwd = real
            ops.AUTO_DEFER = was

    ref, got = grads(False, False), grads(True, True)
    assert set(ref) == set(got)
    top = max(v.abs().max().item() for v in ref.values())
    for n in ref:
        assert (ref[n].float() - got[n].float()).abs().max().item() <= 1e-4 * top, n


def test_second_backward_without_zero_grad_keeps_the_first_pass_gradients():
    """a backward pass that touches FEW parameters (here: the `language` call alone - the text encoder) hands every other
    parameter its `None` back at the end; the next pass without zero_grad then assigns most parameters anew, and GradSession.begin() must not
    zero-fill the whole persistent buffer, because the first pass's gradients are views of it. Both programs (session on / off) must hold the SUM."""
    from tests.golden.variants import HAMT_C1
    from tests.test_hamt_gpu import build_product
    from vln_imagine_amd import dropin, ops, synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors
    cfg = HamtConfig(**HAMT_C1)
    et = EpisodeTensors(synth.HamtEpisode(tag="twice", B=8, L=80, V=37, I=4, T=2, ragged=True), "cuda")

    def program(on):
        was = ops.AUTO_DEFER
        ops.AUTO_DEFER = on
        try:
            ops.reseed(5)
            torch.manual_seed(1)
            m = build_product(cfg, torch.float32).eval()
            w = dropin.wrap_hamt(m, feat_dropout=0.0)
            txt = w("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
            (txt.float() ** 2).mean().backward()                       # pass 1: text-encoder parameters only
            first = {n for n, p in m.named_parameters() if p.grad is not None}
            loss, _ = dropin.hamt_agent_loss(w, et)
            loss.backward()                                            # pass 2, no zero_grad in between
            return first, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.AUTO_DEFER = was

    (f0, g0), (f1, g1) = program(False), program(True)
    assert f0 == f1 and 0 < len(f1) < len(g1) / 2, (len(f0), len(f1), len(g1))      # the case of the finding: most parameters untouched by pass 1
    assert set(g0) == set(g1)
    top = max(v.abs().max().item() for v in g0.values())
    for n in g0:
        d = (g0[n] - g1[n]).abs().max().item()
        assert d <= 1e-4 * top and d <= 2e-3 * max(g0[n].abs().max().item(), 1e-4 * top), (n, n in f1, d, top)


def _graphed_program(on, iters, p_drop, T=3, adam=False, family="hamt", L=80, V=None):
    """iters agent iterations (dropin.*_agent_loss + backward [+ torch.optim.AdamW]) on a wrapper with the per-call graphs on / off."""
    from tests.golden.variants import DUET_C1, HAMT_C1
    from vln_imagine_amd import dropin, graphed, ops, synth
    drops = dict(hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    if family == "hamt":
        from tests.test_hamt_gpu import build_product
        from vln_imagine_amd.hamt.config import HamtConfig
        from vln_imagine_amd.hamt.episode import EpisodeTensors
        cfg = HamtConfig(**HAMT_C1, pred_head_dropout_prob=p_drop, **drops)
        et = EpisodeTensors(synth.HamtEpisode(tag="graphed", B=8, L=L, V=V or 37, I=4, T=T, ragged=True), "cuda")
        wrap, loss_of = dropin.wrap_hamt, dropin.hamt_agent_loss
    else:
        from tests.test_duet_gpu import build_product
        from vln_imagine_amd.duet.config import DuetConfig
        from vln_imagine_amd.duet.episode import DuetEpisodeTensors
        cfg = DuetConfig(**DUET_C1, **drops)
        et = DuetEpisodeTensors(synth.DuetEpisode(tag="graphed", B=8, L=L, V=V or 36, I=4, T=T, ragged=True), "cuda")
        wrap, loss_of = dropin.wrap_duet, dropin.duet_agent_loss
    was = graphed.ENABLED
    graphed.ENABLED = on
    try:
        ops.reseed(9)
        torch.manual_seed(2)
        m = build_product(cfg, torch.bfloat16).train()
        w = wrap(m, feat_dropout=0.0)
        tr = dropin.DropInTrainer(w, et, family, lr=2e-5)
        out = []
        for it in range(iters):
            tr.opt.zero_grad()
            loss, logits = loss_of(w, et)
            logits = logits or []
            loss.backward()
            torch.nn.utils.clip_grad_norm_(tr.params, 40.0)
            out.append((float(loss.detach()), [lg.detach().float().clone() for lg in logits],
                        {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
            if adam:
                tr.opt.step()
            else:
                # the weights move the same way in both programs (an optimizer would feed each program's own summation-order noise back through
                # Adam's normalisation): whether every 16-bit weight copy INSIDE the graphs followed shows in the next iteration
                with torch.no_grad():
                    for i, p in enumerate(tr.params):
                        p.mul_(1.0 + 1e-3 * ((i % 5) - 2))
        return out, dict(graphed.of(m).stats)
    finally:
        graphed.ENABLED = was
        ops.set_seed_base(None)


@pytest.mark.parametrize("family", ["hamt", "duet"])
def test_graphed_wrapper_calls_pad_ragged_shapes_to_buckets(family):
    """graphed.BUCKETS: 73 text tokens, 35 candidate views (HAMT) / maps of 5, 8, 11 nodes (DUET) are padded to 80 / 36 / 8, 8, 16 with the agents' own
    padding values before the captured call and sliced back after it: the outputs have the caller's shapes and, with the gradients, equal the unpadded
    eager calls' to summation-order rounding (padded keys are masked, padded candidates / nodes score -inf)."""
    kw = dict(family=family, L=73, V=35 if family == "hamt" else None)
    (ref, _), (got, stats) = _graphed_program(False, 3, 0.0, **kw), _graphed_program(True, 3, 0.0, **kw)
    assert stats["replayed"] > 0
    for it, ((l0, lg0, g0), (l1, lg1, g1)) in enumerate(zip(ref, got)):
        assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)), (it, l0, l1)
        for a, b in zip(lg0, lg1):
            assert a.shape == b.shape
            fin = torch.isfinite(a)
            assert torch.equal(fin, torch.isfinite(b)) and (a[fin] - b[fin]).abs().max().item() <= 3e-2, it
        assert set(g0) == set(g1)
        top = max(v.abs().max().item() for v in g0.values())
        # (padding changes the launches' row counts, hence tile shapes and summation orders: bf16 noise of a different draw than the equal-shape test's;
        #  the whole gradient agrees to a few percent of its norm, any single entry to 5 % of the largest one)
        num = sum(float((g0[n].double() - g1[n].double()).pow(2).sum()) for n in g0)
        den = sum(float(g0[n].double().pow(2).sum()) for n in g0)
        assert (num / den) ** 0.5 <= 0.05, (it, (num / den) ** 0.5)
        for n in g0:
            assert (g0[n].float() - g1[n].float()).abs().max().item() <= 5e-2 * top, (it, n)


@pytest.mark.parametrize("family", ["hamt", "duet"])
def test_one_autograd_node_per_wrapper_call_equals_the_eager_calls(family):
    """vln_imagine_amd/graphed.py: from the second sighting of a (mode, shapes, occurrence) on, a wrapper call is ONE autograd node that replays
    captured forward / backward hipGraphs. Dropout off: losses, logits and every gradient of four agent iterations with moving weights
    (the 16-bit weight copies inside the graphs must follow) equal the eager calls' to 16-bit summation-order noise."""
    (ref, _), (got, stats) = _graphed_program(False, 4, 0.0, family=family), _graphed_program(True, 4, 0.0, family=family)
    assert stats["captured"] >= 2 * 3 + 2 and stats["replayed"] >= 3 * stats["captured"] - 3 * (2 * 3 + 2), stats
    for it, ((l0, lg0, g0), (l1, lg1, g1)) in enumerate(zip(ref, got)):
        assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)), (it, l0, l1)
        for a, b in zip(lg0, lg1):
            fin = torch.isfinite(a)
            assert torch.equal(fin, torch.isfinite(b)) and (a[fin] - b[fin]).abs().max().item() <= 3e-2, it
        assert set(g0) == set(g1), (it, sorted(set(g0) ^ set(g1))[:5])
        top = max(v.abs().max().item() for v in g0.values())
        for n in g0:
            d = (g0[n].float() - g1[n].float()).abs().max().item()
            assert d <= 2e-2 * top, (it, n, d, top)


def test_graphed_wrapper_calls_draw_new_dropout_masks_every_iteration():
    """In-kernel dropout inside replayed graphs: the seeds are constants of the graphs, the device-resident base advances once per backward pass,
    so two consecutive replayed iterations at the SAME weights (lr = 0 would be needed for equality anyway) differ, and training still converges
    like the eager path: the loss after a few steps stays within the spread of the eager program's."""
    (ref, _), (got, stats) = _graphed_program(False, 5, 0.1, adam=True), _graphed_program(True, 5, 0.1, adam=True)
    assert stats["replayed"] > 0
    l_ref, l_got = [r[0] for r in ref], [g[0] for g in got]
    assert all(abs(a - b) <= 0.15 * abs(a) for a, b in zip(l_ref, l_got)), (l_ref, l_got)
    # replayed iterations 3 and 4 (both from the same graphs): logits differ by more than an optimizer step of 2e-5 would move them
    d = max((a - b)[torch.isfinite(a)].abs().max().item() for a, b in zip(got[3][1], got[4][1]))
    assert d > 1e-3, d


def test_graphed_wrapper_calls_two_rollouts_one_backward_and_state_gradients():
    """The HAMT agent's training iteration is TWO rollouts (teacher-forced, then sampled) and ONE backward of the summed loss (agent_cmt.py:809-832), and its
    critic differentiates the returned states (:700-745): the k-th `visual` call of the second rollout is occurrence T + k of its signature (its own graphs and
    buffers), and an output that carries a gradient only sometimes (states) is served from the same backward graph (zero-filled when absent)."""
    from tests.golden.variants import HAMT_C1
    from tests.test_hamt_gpu import build_product
    from vln_imagine_amd import dropin, graphed, ops, synth
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors
    cfg = HamtConfig(**HAMT_C1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pred_head_dropout_prob=0.0)
    et = EpisodeTensors(synth.HamtEpisode(tag="two", B=8, L=80, V=37, I=4, T=2, ragged=True), "cuda")

    def program(on):
        was = graphed.ENABLED
        graphed.ENABLED = on
        try:
            ops.reseed(3)
            torch.manual_seed(5)
            m = build_product(cfg, torch.bfloat16).train()
            w = dropin.wrap_hamt(m, feat_dropout=0.0)
            out = []
            for it in range(4):
                for p in m.parameters():
                    p.grad = None
                k1, k2 = {}, {}
                l1, _ = dropin.hamt_agent_loss(w, et, keep=k1)
                l2, _ = dropin.hamt_agent_loss(w, et, use_aux=False, keep=k2)
                loss = l1 + 0.5 * l2
                if it % 2 == 1:                                        # every other iteration the critic's term: the states carry a gradient
                    loss = loss + 1e-2 * sum((s.float() ** 2).mean() for s in k2["states"])
                loss.backward()
                out.append((float(loss.detach()), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
                with torch.no_grad():
                    for i, p in enumerate(m.parameters()):
                        p.mul_(1.0 + 1e-3 * ((i % 5) - 2))
            return out, dict(graphed.of(m).stats)
        finally:
            graphed.ENABLED = was
            ops.set_seed_base(None)

    (ref, _), (got, stats) = program(False), program(True)
    assert stats["captured"] >= 2 * (2 * 2 + 2) - 1 and stats["replayed"] >= 2 * stats["captured"], stats      # both rollouts' calls have entries of their own
    for it, ((l0, g0), (l1, g1)) in enumerate(zip(ref, got)):
        assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)), (it, l0, l1)
        assert set(g0) == set(g1), (it, sorted(set(g0) ^ set(g1))[:5])
        top = max(v.abs().max().item() for v in g0.values())
        for n in g0:
            d = (g0[n].float() - g1[n].float()).abs().max().item()
            assert d <= 2e-2 * top, (it, n, d, top)
