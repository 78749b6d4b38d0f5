"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the HAMT NavCMT
hot path as pure functions over a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity is PINNED: tests/test_oracle_hamt.py checks it against tests/golden/hamt_*.npz,
which tests/golden/make_golden_hamt.py produced by running the reference itself.

Every function cites the reference lines it restates; `R:` =
/root/reference/VLN-HAMT/finetune_src/models/vilmodel_cmt.py.
"""
import math

import torch
import torch.nn.functional as F

NEG = -10000.0


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x, eps=1e-12):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def gelu_erf(x):                                    # R:27-33
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def ext_mask(m):                                    # R:1010-1012 (1-m)*-10000, [B,1,1,S]
    return (1.0 - m[:, None, None, :].float()) * NEG


def mha(sd, p, q_in, kv_in, add_mask, nh=12):       # R:100-134 / R:326-353
    B, Sq, H = q_in.shape
    Sk = kv_in.shape[1]
    dh = H // nh
    q = _lin(sd, p + ".query", q_in).view(B, Sq, nh, dh).transpose(1, 2)
    k = _lin(sd, p + ".key", kv_in).view(B, Sk, nh, dh).transpose(1, 2)
    v = _lin(sd, p + ".value", kv_in).view(B, Sk, nh, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    pr = torch.softmax(s, -1)
    return (pr @ v).transpose(1, 2).reshape(B, Sq, H)


def att_probs(sd, p, q_in, kv_in, add_mask, nh=12):  # softmax of the raw scores an attention returns (R:118-124 with output_attentions)
    B, Sq, H = q_in.shape
    Sk = kv_in.shape[1]
    dh = H // nh
    q = _lin(sd, p + ".query", q_in).view(B, Sq, nh, dh).transpose(1, 2)
    k = _lin(sd, p + ".key", kv_in).view(B, Sk, nh, dh).transpose(1, 2)
    return torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + add_mask, -1)


def self_output(sd, p, h, resid):                   # R:144-148 / R:186-190
    return _ln(sd, p + ".LayerNorm", _lin(sd, p + ".dense", h) + resid)


def bert_attention(sd, p, x, add_mask):             # R:157-161
    return self_output(sd, p + ".output", mha(sd, p + ".self", x, x, add_mask), x)


def ffn(sd, p_inter, p_out, x):                     # R:173-176,186-190
    return self_output(sd, p_out, gelu_erf(_lin(sd, p_inter + ".dense", x)), x)


def bert_layer(sd, p, x, add_mask):                 # R:200-206
    a = bert_attention(sd, p + ".attention", x, add_mask)
    return ffn(sd, p + ".intermediate", p + ".output", a)


def x_attention(sd, p, x, ctx, ctx_mask):           # R:361-364
    return self_output(sd, p + ".output", mha(sd, p + ".att", x, ctx, ctx_mask), x)


def lxrt_layer(sd, p, lang, lang_mask, visn, visn_mask):   # R:423-445
    # cross attention, shared weights, both directions read the PRE-update inputs (R:385-397)
    l1 = x_attention(sd, p + ".visual_attention", lang, visn, visn_mask)
    v1 = x_attention(sd, p + ".visual_attention", visn, lang, lang_mask)
    l2 = bert_attention(sd, p + ".lang_self_att", l1, lang_mask)      # R:399-407
    v2 = bert_attention(sd, p + ".visn_self_att", v1, visn_mask)
    l3 = ffn(sd, p + ".lang_inter", p + ".lang_output", l2)            # R:409-421
    v3 = ffn(sd, p + ".visn_inter", p + ".visn_output", v2)
    return l3, v3


def lxrt_layer_no_lang_ca(sd, p, lang, lang_mask, visn, visn_mask):   # R:385-421 with self.no_lang_ca: the vision stream only
    v1 = x_attention(sd, p + ".visual_attention", visn, lang, lang_mask)
    v2 = bert_attention(sd, p + ".visn_self_att", v1, visn_mask)
    return ffn(sd, p + ".visn_inter", p + ".visn_output", v2)


def lxrt_layer_probs(sd, p, lang, lang_mask, visn, visn_mask):
    """The four visualisation maps of a layer (R:391,393,438,439): cross-attention pair on its inputs, self-attention pair on the
    cross-attention outputs."""
    xa = p + ".visual_attention"
    l1 = x_attention(sd, xa, lang, visn, visn_mask)
    v1 = x_attention(sd, xa, visn, lang, lang_mask)
    return (att_probs(sd, xa + ".att", lang, visn, visn_mask), att_probs(sd, xa + ".att", visn, lang, lang_mask),
            att_probs(sd, p + ".lang_self_att.self", l1, l1, lang_mask), att_probs(sd, p + ".visn_self_att.self", v1, v1, visn_mask))


class HamtOracle:
    def __init__(self, cfg, sd):
        self.cfg, self.sd = cfg, sd

    # ---- embeddings -------------------------------------------------------------
    def bert_embeddings(self, ids):                 # R:58-73
        sd = self.sd
        L = ids.shape[1]
        e = (sd["embeddings.word_embeddings.weight"][ids]
             + sd["embeddings.position_embeddings.weight"][:L][None]
             + sd["embeddings.token_type_embeddings.weight"][0][None, None])
        return _ln(sd, "embeddings.LayerNorm", e)

    def image_embeddings(self, img, ang, nav_types):  # R:535-544 with type id 1 (R:1074-1077)
        sd, p = self.sd, "img_embeddings"
        e = (_ln(sd, p + ".img_layer_norm", _lin(sd, p + ".img_linear", img))
             + _ln(sd, p + ".ang_layer_norm", _lin(sd, p + ".ang_linear", ang))
             + sd["embeddings.token_type_embeddings.weight"][1][None, None]
             + sd[p + ".nav_type_embedding.weight"][nav_types])
        return _ln(sd, p + ".layer_norm", e)

    def history_embeddings(self, img, ang, pos_ids, pano_img, pano_ang):   # R:576-618
        sd, p = self.sd, "hist_embeddings"
        typ = sd[p + ".type_embedding.weight"][0][None]
        if img is None:                              # CLS path R:592-595
            return _ln(sd, p + ".layer_norm", sd[p + ".cls_token"][:, 0] + typ)
        e = (_ln(sd, p + ".img_layer_norm", _lin(sd, p + ".img_linear", img))
             + _ln(sd, p + ".ang_layer_norm", _lin(sd, p + ".ang_linear", ang))
             + sd[p + ".position_embeddings.weight"][pos_ids] + typ)
        if self.cfg.hist_enc_pano:                   # R:603-614
            pe = (_ln(sd, p + ".pano_img_layer_norm", _lin(sd, p + ".pano_img_linear", pano_img))
                  + _ln(sd, p + ".pano_ang_layer_norm", _lin(sd, p + ".pano_ang_linear", pano_ang)))
            for i in range(self.cfg.num_h_pano_layers):
                pe = bert_layer(sd, f"{p}.pano_encoder.layer.{i}", pe, None)   # mask is all zeros
            e = e + pe.mean(1)
        return _ln(sd, p + ".layer_norm", e)

    def imagine_embeddings(self, feats, masks):
        sd, p = self.sd, "imagine_embeddings"
        typ = sd[p + ".type_embedding.weight"][0][None, None]
        if self.cfg.bypass_imag_encoder:             # R:625-631
            return feats + typ
        n = feats.shape[1]                           # R:659-703
        assert n < self.cfg.max_imagination_len
        x = feats + sd[p + ".position_embeddings.weight"][:n][None] + typ
        x = _ln(sd, p + ".pano_img_layer_norm", _lin(sd, p + ".pano_img_linear", x))
        m = ext_mask(masks)
        for i in range(self.cfg.num_h_pano_layers):
            x = bert_layer(sd, f"{p}.pano_encoder.layer.{i}", x, m)
        return _ln(sd, p + ".layer_norm", x)

    # ---- aux head ---------------------------------------------------------------
    def _proj(self, x):                              # R:714-728 (dropout = identity in eval)
        sd, p = self.sd, "contrastive_alignment_model.image_proj"
        return F.linear(F.relu(F.linear(F.relu(F.linear(x, sd[p + ".fc1.weight"])),
                                        sd[p + ".fc2.weight"])), sd[p + ".fc3.weight"])

    def align(self, txt, txt_masks, img, img_masks, segs, flags, nps):
        """R:737-790 (cosine) and R:866-950 (InfoNCE / margin). Out-of-place scatter:
        same forward values as the reference's in-place write (R:781), gradient as intended."""
        typ = self.cfg.aux_loss_type
        B = img.shape[0]
        np_means = {}                                # R:876-898: every noun phrase mean, flag-True slots only
        if typ != "cosine":
            for b in range(B):
                np_means[b] = []
                for i, lst in enumerate(nps[b]):
                    if flags[b][i] != "True":
                        continue
                    for (s, e) in lst:
                        np_means[b].append(txt[b, s:e + 1].mean(0))
        out_rows, losses = {}, []
        for b in range(B):
            assert len(flags[b]) == len(segs[b]) == len(nps[b])
            for i in range(len(flags[b])):
                if flags[b][i] != "True":
                    continue
                assert bool(img_masks[b, i])
                proj = self._proj(img[b, i])
                toks = []
                for (s, e) in nps[b][i]:
                    assert s >= segs[b][i][0] and e <= segs[b][i][1]
                    assert bool(txt_masks[b, s:e + 1].all())
                    toks.append(txt[b, s:e + 1])
                if len(nps[b][i]) == 0:
                    continue                         # R:777: MLP ran, nothing written or scored
                mean = torch.cat(toks, 0).mean(0)
                out_rows[(b, i)] = proj
                if typ == "cosine":
                    losses.append(1 - F.cosine_similarity(proj, mean, dim=-1))
                else:
                    negs = [m for bb, ms in np_means.items() if bb != b for m in ms]
                    allt = torch.stack([mean] + negs, 0)
                    sims = F.cosine_similarity(proj[None], allt)
                    if typ == "contrastive-InfoNCE":        # R:793-823
                        losses.append(F.cross_entropy((sims / self.cfg.infonce_temperature)[None],
                                                      torch.zeros(1, dtype=torch.long)))
                    else:                                    # R:825-856
                        pos = sims[0]
                        losses.append((1 - pos) + F.relu(self.cfg.contrastive_margin_value + sims[1:] - pos).mean())
        if out_rows:
            sel = torch.zeros(img.shape[:2], dtype=torch.bool)
            rows = torch.zeros_like(img)
            for (b, i), pr in out_rows.items():
                sel[b, i] = True
            rows = torch.stack([torch.stack([out_rows.get((b, i), img[b, i]) for i in range(img.shape[1])])
                                for b in range(B)])
            new_img = torch.where(sel[..., None], rows, img)
        else:
            new_img = img
        loss = torch.stack(losses).mean() if losses else 0
        return loss, new_img

    # ---- mode dispatch (R:999-1205) ----------------------------------------------
    def __call__(self, mode, **kw):
        cfg, sd = self.cfg, self.sd
        if mode == "language":                       # R:1008-1030
            m = ext_mask(kw["txt_masks"])
            x = self.bert_embeddings(kw["txt_ids"])
            for i in range(cfg.num_l_layers):
                x = bert_layer(sd, f"encoder.layer.{i}", x, m)
            if cfg.fix_lang_embedding:
                x = x.detach()
            if cfg.no_lang_ca:                       # R:1022-1029: every x-layer's language self-attention + FFN on the SAME text states
                outs = [x]                           # (the loop never feeds one layer's output to the next)
                for i in range(cfg.num_x_layers):
                    p = f"encoder.x_layers.{i}"
                    outs.append(ffn(sd, p + ".lang_inter", p + ".lang_output", bert_attention(sd, p + ".lang_self_att", x, m)))
                return outs
            return x
        if mode == "history":                        # R:1033-1038
            h = self.history_embeddings(kw.get("hist_img_feats"), kw.get("hist_ang_feats"),
                                        kw.get("ob_step_ids"), kw.get("hist_pano_img_feats"),
                                        kw.get("hist_pano_ang_feats"))
            return h.detach() if cfg.fix_hist_embedding else h
        if mode == "imagine":                        # R:1040-1048
            e = self.imagine_embeddings(kw["imagine_pano_img_feats"], kw.get("imagine_masks"))
            return e.detach() if cfg.fix_imagine_embeds else e
        if mode == "align_with_contrastive_loss":    # R:1050-1053
            return self.align(kw["align_txt_embeds"], kw["txt_masks"], kw["align_imagine_embeds"],
                              kw["imagine_masks"], kw["sub_instr_segs"], kw["sub_instr_imag_flag"],
                              kw["noun_phrase_segs"])
        assert mode == "visual"                      # R:1056-1205
        hist, txt = kw["hist_embeds"], kw["txt_embeds"]
        hm, om, tm = ext_mask(kw["hist_masks"]), ext_mask(kw["ob_masks"]), ext_mask(kw["txt_masks"])
        for i in range(cfg.num_h_layers):            # R:1064-1067 temporal history transformer
            hist = bert_layer(sd, f"encoder.h_layers.{i}", hist, hm)
        ob = self.image_embeddings(kw["ob_img_feats"], kw["ob_ang_feats"], kw["ob_nav_types"])
        for i in range(cfg.num_r_layers):            # R:1079-1081
            ob = bert_layer(sd, f"encoder.r_layers.{i}", ob, om)
        if cfg.fix_obs_embedding:
            ob = ob.detach()
        txt_list = txt if isinstance(txt, list) else None     # no_lang_ca: [text, x-layer 0's text states, ...] (R:1097-1100)
        if txt_list is not None:
            assert not (cfg.imagine_enc_pano and cfg.concat_imagine_with == "language"), "R:1110 concatenates a list: the reference raises"
            txt = txt_list[0]
        nh, nt, no = hist.shape[1], txt.shape[1], ob.shape[1]
        visn, vm = torch.cat([hist, ob], 1), torch.cat([hm, om], -1)
        lang, lm = txt, tm
        img = kw.get("imagine_embeds")
        if cfg.imagine_enc_pano:
            im = ext_mask(kw["imagine_masks"])
            if cfg.concat_imagine_with == "visual":  # R:1106-1108
                visn, vm = torch.cat([visn, img], 1), torch.cat([vm, im], -1)
            else:                                    # R:1109-1112
                lang, lm = torch.cat([lang, img], 1), torch.cat([lm, im], -1)
        cross_probs, self_probs = [], []
        for i in range(cfg.num_x_layers):
            if kw.get("return_cross_attention_probs"):   # R:1128-1153
                lq, vq, ls, vs = lxrt_layer_probs(sd, f"encoder.x_layers.{i}", lang, lm, visn, vm)
                cross_probs.append((lq, vq)); self_probs.append((ls, vs))
            if cfg.no_lang_ca:                       # R:1118-1127 / 1157-1164 with R:385-421 under the flag: layer i reads entry i of the
                lang = txt_list[i]                   # list, only the vision stream is updated, the language input is returned unchanged
                visn = lxrt_layer_no_lang_ca(sd, f"encoder.x_layers.{i}", lang, lm, visn, vm)
            else:
                lang, visn = lxrt_layer(sd, f"encoder.x_layers.{i}", lang, lm, visn, vm)
        hist_o, ob_o, txt_o = visn[:, :nh], visn[:, nh:nh + no], lang[:, :nt]
        if cfg.imagine_enc_pano:
            img_o = visn[:, nh + no:] if cfg.concat_imagine_with == "visual" else lang[:, nt:]
        tok = cfg.act_pred_token                     # R:1190-1199
        if cfg.no_lang_ca:                           # R:1187-1188
            f = ob_o
        elif tok == "ob_txt":
            f = ob_o * txt_o[:, :1]
        elif tok == "ob":
            f = ob_o
        elif tok == "ob_hist":
            f = ob_o * hist_o[:, :1]
        elif tok == "ob_txt_hist":
            f = ob_o * (txt_o[:, :1] + hist_o[:, :1])
        else:
            f = ob_o * (txt_o[:, :1] + img_o.mean(1, keepdim=True))
        h = _ln(sd, "next_action.net.2", F.relu(_lin(sd, "next_action.net.0", f)))   # R:956-960
        logits = _lin(sd, "next_action.net.4", h).squeeze(-1)
        logits = logits.masked_fill(kw["ob_nav_types"] == 0, -float("inf"))           # R:1200
        if kw.get("return_cross_attention_probs"):
            return logits, txt_o, hist_o, ob_o, cross_probs, self_probs
        return logits, txt_o, hist_o, ob_o
