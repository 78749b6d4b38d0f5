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
