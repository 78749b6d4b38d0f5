"""ORACLE (test infrastructure, not product): CPU fp32 restatement of DUET's GlocalTextPathNavCMT hot path as pure
functions over a state_dict. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
Parity is PINNED by tests/test_oracle_duet.py against tests/golden/duet_*.npz (made by running the reference).

`D:` = /root/reference/VLN-DUET/map_nav_src/models/vilmodel.py, `T:` = .../models/transformer.py,
`O:` = .../models/ops.py.
"""
import torch
import torch.nn.functional as F

from oracle.hamt_oracle import HamtOracle, _lin, _ln, bert_attention, bert_layer, ext_mask, ffn, x_attention


def graph_x_layer(sd, p, lang, lang_mask, visn, visn_mask, sprels):          # D:384-399
    v = x_attention(sd, p + ".visual_attention", visn, lang, lang_mask)
    m = visn_mask if sprels is None else visn_mask + sprels
    v = bert_attention(sd, p + ".visn_self_att", v, m)
    return ffn(sd, p + ".visn_inter", p + ".visn_output", v)


def prenorm_layer(sd, p, x, key_pad):                                          # T:170-182 (forward_pre), eps 1e-5
    B, S, H = x.shape
    nh, dh = 12, H // 12
    h = F.layer_norm(x, (H,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
    q, k, v = [t.view(B, S, nh, dh).transpose(1, 2) for t in qkv.split(H, -1)]
    s = q @ k.transpose(-1, -2) / dh ** 0.5
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, H)
    x = x + _lin(sd, p + ".self_attn.out_proj", a)
    h = F.layer_norm(x, (H,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return x + _lin(sd, p + ".linear2", F.gelu(_lin(sd, p + ".linear1", h)))


def cls_head(sd, p, x):                                                        # D:1009-1020
    return _lin(sd, p + ".net.3", _ln(sd, p + ".net.2", F.relu(_lin(sd, p + ".net.0", x))))


class DuetOracle:
    def __init__(self, cfg, sd):
        self.cfg, self.sd = cfg, sd
        self._h = HamtOracle(cfg, sd)       # shares BertEmbeddings and the aux head (same math: D:591-655 == R:737-790)

    def panorama(self, b):                                                     # D:1087-1131
        sd, p = self.sd, "img_embeddings"
        img = _ln(sd, p + ".img_layer_norm", _lin(sd, p + ".img_linear", b["view_img_fts"]))
        lens = b["view_lens"]
        if b.get("obj_img_fts") is not None:                                   # D:1096-1114 (REVERIE / SOON objects)
            q = "obj" if (p + ".obj_linear.weight") in sd else "img"
            obj = _ln(sd, f"{p}.{q}_layer_norm", _lin(sd, f"{p}.{q}_linear", b["obj_img_fts"]))
            per = [torch.cat([img[i, :int(lens[i])], obj[i, :int(b["obj_lens"][i])]], 0) for i in range(img.shape[0])]
            S = max(x.shape[0] for x in per)
            img = torch.stack([torch.cat([x, x.new_zeros(S - x.shape[0], x.shape[1])], 0) for x in per])      # O:46-68
            lens = lens + b["obj_lens"]
        e = (img
             + _ln(sd, p + ".loc_layer_norm", _lin(sd, p + ".loc_linear", b["loc_fts"]))
             + sd[p + ".nav_type_embedding.weight"][b["nav_types"]]
             + sd["embeddings.token_type_embeddings.weight"][1][None, None])
        e = _ln(sd, p + ".layer_norm", e)
        masks = torch.arange(e.shape[1])[None, :] < lens[:, None]             # O:36-44
        for i in range(self.cfg.num_pano_layers):
            e = prenorm_layer(sd, f"{p}.pano_encoder.layers.{i}", e, ~masks)
        if self.cfg.num_pano_layers > 0:
            e = _ln(sd, p + ".pano_encoder.norm", e)
        return e, masks

    def navigation(self, b):                                                   # D:1133-1235
        sd, cfg = self.sd, self.cfg
        txt, tm = b["txt_embeds"], b["txt_masks"]
        g = (b["gmap_img_embeds"] + sd["global_encoder.gmap_step_embeddings.weight"][b["gmap_step_ids"]]
             + _ln(sd, "global_encoder.gmap_pos_embeddings.1", _lin(sd, "global_encoder.gmap_pos_embeddings.0", b["gmap_pos_fts"])))
        sprels = None
        if cfg.graph_sprels:                                                   # D:1145-1147
            sprels = (b["gmap_pair_dists"] * sd["global_encoder.sprel_linear.weight"][0, 0]
                      + sd["global_encoder.sprel_linear.bias"][0])[:, None]
        v = b["vp_img_embeds"] + _ln(sd, "local_encoder.vp_pos_embeddings.1",
                                     _lin(sd, "local_encoder.vp_pos_embeddings.0", b["vp_pos_fts"]))
        if cfg.imagine_enc_pano and cfg.concat_imagine_with == "language":     # D:1157-1158
            txt = torch.cat([txt, b["imagine_embeds"]], 1)
            tm = torch.cat([tm, b["imagine_masks"]], 1)
        lm, gm, vm = ext_mask(tm), ext_mask(b["gmap_masks"]), ext_mask(b["vp_masks"])
        for i in range(cfg.num_x_layers):
            g = graph_x_layer(sd, f"global_encoder.encoder.x_layers.{i}", txt, lm, g, gm, sprels)
        for i in range(cfg.num_x_layers):
            v = graph_x_layer(sd, f"local_encoder.encoder.x_layers.{i}", txt, lm, v, vm, None)
        fuse = torch.sigmoid(cls_head(sd, "sap_fuse_linear", torch.cat([g[:, 0], v[:, 0]], 1))) if cfg.glocal_fuse else 0.5
        gl = cls_head(sd, "global_sap_head", g).squeeze(2) * fuse
        gl = gl.masked_fill(b["gmap_visited_masks"], -float("inf")).masked_fill(~b["gmap_masks"], -float("inf"))
        ll = cls_head(sd, "local_sap_head", v).squeeze(2) * (1 - fuse)
        ll = ll.masked_fill(~b["vp_nav_masks"], -float("inf"))
        rows = []
        for i in range(gl.shape[0]):                                           # fusion loop D:1200-1217
            visited = {vp for vp, m in zip(b["gmap_vpids"][i], b["gmap_visited_masks"][i].tolist()) if m}
            tmp, bw = {}, 0
            for j, c in enumerate(b["vp_cand_vpids"][i]):
                if j > 0:
                    if c in visited:
                        bw = bw + ll[i, j]
                    else:
                        tmp[c] = ll[i, j]
            row = [gl[i, 0] + ll[i, 0]]
            for j in range(1, gl.shape[1]):
                vp = b["gmap_vpids"][i][j] if j < len(b["gmap_vpids"][i]) else None
                if j < len(b["gmap_vpids"][i]) and vp not in visited:
                    row.append(gl[i, j] + (tmp[vp] if vp in tmp else bw))
                else:
                    row.append(gl[i, j])
            rows.append(torch.stack(row))
        fused = torch.stack(rows)
        obj = None
        if b.get("vp_obj_masks") is not None:                                  # D:1220-1225
            obj = cls_head(sd, "og_head", v).squeeze(2).masked_fill(~b["vp_obj_masks"], -float("inf"))
        return {"gmap_embeds": g, "vp_embeds": v, "global_logits": gl, "local_logits": ll, "fused_logits": fused,
                "obj_logits": obj}

    def align_reverie(self, txt, txt_masks, img, img_masks):                   # D:781-888
        """Imagination 0 of every sample vs the mean of all valid instruction tokens; negatives = other samples' means."""
        cfg, typ, B = self.cfg, self.cfg.aux_loss_type, img.shape[0]
        means = {b: txt[b][txt_masks[b]].mean(0) for b in range(B) if bool(txt_masks[b].any())}
        rows, losses = [], []
        for b in range(B):
            assert bool(img_masks[b].reshape(-1)[0])
            proj = self._h._proj(img[b, 0])
            if b not in means:
                rows.append(img[b])
                continue
            rows.append(torch.cat([proj[None], img[b, 1:]], 0))
            if typ == "cosine":
                losses.append(1 - F.cosine_similarity(proj, means[b], dim=-1))
            else:
                allt = torch.stack([means[b]] + [m for bb, m in means.items() if bb != b], 0)
                sims = F.cosine_similarity(proj[None], allt)
                if typ == "contrastive-InfoNCE":                               # D:657-687
                    losses.append(F.cross_entropy((sims / cfg.infonce_temperature)[None], torch.zeros(1, dtype=torch.long)))
                else:                                                          # D:890-921
                    losses.append((1 - sims[0]) + F.relu(cfg.contrastive_margin_value + sims[1:] - sims[0]).mean())
        return (torch.stack(losses).mean() if losses else 0), torch.stack(rows)

    def __call__(self, mode, b):
        cfg, sd = self.cfg, self.sd
        if mode == "language":                                                 # D:1075-1079, 427-434
            x = self._h.bert_embeddings(b["txt_ids"])
            m = ext_mask(b["txt_masks"])
            for i in range(cfg.num_l_layers):
                x = bert_layer(sd, f"lang_encoder.layer.{i}", x, m)
            return x if cfg.update_lang_bert else x.detach()
        if mode == "imagine":                                                  # D:1081-1085, 562-573
            return b["imagine_feats"] + sd["imagine_embeddings.type_embedding.weight"][0][None, None]
        if mode == "align_with_contrastive_loss":                              # D:1247-1263
            txt = b["align_txt_embeds"].detach() if cfg.fix_lang_inside_cosine_model else b["align_txt_embeds"]
            if cfg.dataset == "reverie":
                return self.align_reverie(txt, b["txt_masks"], b["align_imagine_embeds"], b["imagine_masks"])
            return self._h.align(txt, b["txt_masks"], b["align_imagine_embeds"], b["imagine_masks"],
                                 b["sub_instr_segs"], b["sub_instr_imag_flag"], b["noun_phrase_segs"])
        if mode == "panorama":
            return self.panorama(b)
        if mode == "navigation":
            return self.navigation(b)
        raise NotImplementedError("wrong mode: %s" % mode)
