"""MI355X-native DUET `GlocalTextPathNavCMT` (drop-in for VLN-DUET/map_nav_src/models/vilmodel.py:1022-1288).

Same forward(mode, batch) contract, output dict and state_dict keys as the reference. Modules are parameter
holders; arithmetic runs in the fused HIP sublayer operators of vln_imagine_amd.ops:
  language   BertEmbeddings (sum+LN kernel) + 9 fused BertLayers
  panorama   LN(Linear(img)) + LN(Linear(loc, K=7)) + nav-type + token-type -> LN -> 2 PRE-norm encoder layers
             (packed in_proj, -inf key padding, eps 1e-5; transformer.py:170-182) -> LN(1e-12)
  navigation per branch 4 x [cross-attention visn<-text|imagination, self-attention (+ graph_sprels bias on the
             global map), FFN]; SAP heads, dynamic fusion; the reference's per-element Python fusion loop
             (:1200-1217) becomes index tensors built once on the host + three scatter/gather ops.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from vln_imagine_amd import ops
from vln_imagine_amd.hamt.models.vilmodel_cmt import (HID_EPS, AlignWithContrastiveLoss, BertAttention, BertEmbeddings,
                                                     BertIntermediate, BertLayer, BertOutput, BertXAttention,
                                                     BypassImagineEmbeddings, _att, _drop, _ffn)
from .transformer import TransformerEncoder


class AlignWithContrastiveLossReverie(AlignWithContrastiveLoss):
    """REVERIE whole-instruction alignment (reference vilmodel.py:781-888, both the cosine and the negative-sample classes):
    imagination slot 0 of every sample is projected and pulled towards the mean of ALL valid instruction tokens; negatives
    (InfoNCE / margin) are the other samples' instruction means. Same kernels as the R2R head, a different index plan."""

    def _index_plan(self, txt_masks, imagine_masks, sub_instr_segs, sub_instr_imag_flag, noun_phrase_segs, B, L, I, typ, dev):
        key = ("reverie", id(txt_masks), id(imagine_masks), B, L, I, typ)
        hit = self._plans.get(key)
        if hit is not None:
            return hit[0]
        tm, im = txt_masks.cpu(), imagine_masks.cpu()
        mlp_rows, scored, seg_off, tok_rows, owner = [], [], [0], [], []
        for b in range(B):
            assert bool(im[b].reshape(-1)[0]), "Imagine embeds is not valid where embedding addition is being applied."
            mlp_rows.append(b * I)
            toks = [b * L + t for t in range(L) if bool(tm[b, t])]
            if toks:
                scored.append(b)
                tok_rows.extend(toks)
                seg_off.append(len(tok_rows))
                owner.append(b)
        plan = None
        if scored:
            it = lambda v, d=torch.int32: torch.tensor(v, dtype=d, device=dev)
            neg = (it(seg_off), it(tok_rows), it(owner, torch.long)) if typ != "cosine" else (None, None, None)
            plan = (it(mlp_rows, torch.long), it(scored, torch.long), it(seg_off), it(tok_rows)) + neg
        if len(self._plans) >= 16:
            self._plans.clear()
        self._plans[key] = (plan, (txt_masks, imagine_masks))
        return plan


AlignWithContrastiveLossWithNegativeSamplesReverie = AlignWithContrastiveLossReverie


FUSED_EMBED = os.environ.get("VLNI_FUSED_EMBED", "1") == "1"     # A/B switch (round 5): panorama / map-node / viewpoint embeddings through ops.embed_combine
CACHE_TEXT_KV = os.environ.get("VLNI_CACHE_TEXT_KV", "1") == "1"
DUAL_BRANCHES = os.environ.get("VLNI_DUET_DUAL", "1") == "1"      # global + local encoder layers as dual-problem launches


class GraphLXRTXLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        if c.use_lang2visn_attn:
            raise NotImplementedError("use_lang2visn_attn is False in every DUET fine-tuning run (vlnbert_init.py:57)")
        self.visn_self_att = BertAttention(c)
        self.visn_inter = BertIntermediate(c)
        self.visn_output = BertOutput(c)
        self.visual_attention = BertXAttention(c)

    def forward(self, lang, lang_mask, visn, visn_mask, graph_sprels=None, kv=None):
        if kv is not None:      # text-side K/V projected once per episode (CrossmodalEncoder caches it)
            visn = ops.xatt_q_block(visn, kv, lang_mask, _att(self.visual_attention), drop=_drop(self.visual_attention))
        else:
            visn = ops.xatt_block(visn, lang, lang_mask, _att(self.visual_attention), drop=_drop(self.visual_attention))
        visn = self.visn_self_att(visn, visn_mask, bias=graph_sprels)
        return ops.ffn_block(visn, _ffn(self.visn_inter, self.visn_output), drop=_drop(self.visn_output))


class LanguageEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.update_lang_bert = c.update_lang_bert
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c.num_l_layers)])
        if not c.update_lang_bert:
            for p in self.layer.parameters():
                p.requires_grad = False


class CrossmodalEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.x_layers = nn.ModuleList([GraphLXRTXLayer(c) for _ in range(c.num_x_layers)])

    def forward(self, txt, txt_add_mask, visn, visn_add_mask, graph_sprels=None, kvs=None):
        for i, l in enumerate(self.x_layers):
            visn = l(txt, txt_add_mask, visn, visn_add_mask, graph_sprels, kv=None if kvs is None else kvs[i])
        return visn

    def project_context(self, txt):
        """K/V projections of the (step-invariant) text + imagination context for every layer."""
        return [ops.kv_proj(txt, _att(l.visual_attention)) for l in self.x_layers]


class ImageEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        h = c.hidden_size
        self.img_linear = nn.Linear(c.image_feat_size, h)
        self.img_layer_norm = nn.LayerNorm(h, eps=HID_EPS)
        self.loc_linear = nn.Linear(c.angle_feat_size + 3, h)
        self.loc_layer_norm = nn.LayerNorm(h, eps=HID_EPS)
        if c.obj_feat_size > 0 and c.obj_feat_size != c.image_feat_size:      # reference :464-468
            self.obj_linear = nn.Linear(c.obj_feat_size, h)
            self.obj_layer_norm = nn.LayerNorm(h, eps=HID_EPS)
        else:
            self.obj_linear = self.obj_layer_norm = None
        self.nav_type_embedding = nn.Embedding(3, h)                          # 0 non-navigable, 1 navigable, 2 object
        self.layer_norm = nn.LayerNorm(h, eps=HID_EPS)
        self.pano_encoder = TransformerEncoder(c, c.num_pano_layers) if c.num_pano_layers > 0 else None


class _PosEmbed(nn.Sequential):
    """Sequential(Linear(K, 768), LayerNorm) holder, K <= 16 (7-d / 14-d position features)."""

    def __init__(self, k, h):
        super().__init__(nn.Linear(k, h), nn.LayerNorm(h, eps=HID_EPS))

    def embed(self, x, dt):
        return ops.layer_norm(ops.smallk_linear(x, self[0].weight, self[0].bias, dt), self[1].weight, self[1].bias, HID_EPS)


class LocalVPEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.vp_pos_embeddings = _PosEmbed(c.angle_feat_size * 2 + 6, c.hidden_size)
        self.encoder = CrossmodalEncoder(c)


class GlobalMapEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gmap_pos_embeddings = _PosEmbed(c.angle_feat_size + 3, c.hidden_size)
        self.gmap_step_embeddings = nn.Embedding(c.max_action_steps, c.hidden_size)
        self.encoder = CrossmodalEncoder(c)
        self.sprel_linear = nn.Linear(1, 1) if c.graph_sprels else None


class ClsPrediction(nn.Module):
    def __init__(self, hidden, input_size=None):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(input_size or hidden, hidden), nn.ReLU(), nn.LayerNorm(hidden, eps=HID_EPS),
                                 nn.Linear(hidden, 1))

    def forward(self, x, neg_inf_mask=None):
        n = self.net
        if FUSED_EMBED:              # LayerNorm + Linear(768 -> 1) (+ masked_fill): one launch (ops.ln_rowdot)
            return ops.ln_rowdot(ops.linear(x, n[0].weight, n[0].bias, act=2), n[2].weight, n[2].bias, n[3].weight, n[3].bias, neg_inf_mask, eps=HID_EPS)
        h = ops.layer_norm(ops.linear(x, n[0].weight, n[0].bias, act=2), n[2].weight, n[2].bias, HID_EPS)
        return ops.row_dot(h, n[3].weight, n[3].bias, neg_inf_mask)


def _cfg(config):
    from vln_imagine_amd.duet.config import DuetConfig
    if isinstance(config, DuetConfig):
        return config
    d = config.to_dict() if hasattr(config, "to_dict") else dict(config.__dict__)
    known = set(DuetConfig().__dict__)
    return DuetConfig(**{k: v for k, v in d.items() if k in known})


class GlocalTextPathNavCMT(nn.Module):
    def __init__(self, config):
        super().__init__()
        c = self.config = _cfg(config)
        self.embeddings = BertEmbeddings(c)
        self.lang_encoder = LanguageEncoder(c)
        self.img_embeddings = ImageEmbeddings(c)
        self.local_encoder = LocalVPEncoder(c)
        self.global_encoder = GlobalMapEncoder(c)
        self.global_sap_head = ClsPrediction(c.hidden_size)
        self.local_sap_head = ClsPrediction(c.hidden_size)
        self.sap_fuse_linear = ClsPrediction(c.hidden_size, input_size=c.hidden_size * 2) if c.glocal_fuse else None
        if c.obj_feat_size > 0:
            self.og_head = ClsPrediction(c.hidden_size)                         # object grounding, reference :1039-1040
        if c.imagine_enc_pano:
            if c.bypass_imag_encoder:
                self.imagine_embeddings = BypassImagineEmbeddings(c)
            if c.use_cosine_aux_loss or c.no_loss_test:
                self.contrastive_alignment_model = (AlignWithContrastiveLossReverie if c.dataset == "reverie"
                                                    else AlignWithContrastiveLoss)(c)
        from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
        self.apply(NavCMT._init_weights)
        if c.fix_lang_embedding or c.fix_local_branch:
            for m in (self.embeddings, self.lang_encoder):
                for p in m.parameters():
                    p.requires_grad = False
        if c.fix_pano_embedding or c.fix_local_branch:
            for p in self.img_embeddings.parameters():
                p.requires_grad = False
        if c.fix_local_branch:                       # reference :1066-1073 (self.og_head exists with object features only; the reference
            for m in (self.local_encoder, self.local_sap_head, self.og_head):      # raises AttributeError here without them, and so does this)
                for p in m.parameters():
                    p.requires_grad = False
        self.compute_dtype = torch.bfloat16 if os.environ.get("VLNI_DTYPE", "fp32").lower() in ("bf16", "bfloat16") \
            else torch.float32
        self._kv_cache = None
        self._fuse_plans = {}

    def _drop_kv_cache(self, _grad=None):
        self._kv_cache = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None):
        m = cls(config)
        if state_dict:
            m.load_state_dict({(k[5:] if k.startswith("bert.") else k): v for k, v in state_dict.items()}, strict=False)
        return m

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.compute_dtype = dtype
        return self

    @property
    def device(self):
        return self.global_sap_head.net[0].weight.device

    # ---- modes ---------------------------------------------------------------------------------
    def forward_text(self, txt_ids, txt_masks):
        dt, e = self.compute_dtype, self.embeddings
        B, L = txt_ids.shape
        pos = torch.arange(L, device=txt_ids.device).repeat(B)
        srcs = [(e.word_embeddings.weight, "gather", txt_ids.reshape(-1).contiguous()),
                (e.position_embeddings.weight, "gather", pos), (e.token_type_embeddings.weight[0], "bcast", None)]
        x = ops.sum_layer_norm(srcs, e.LayerNorm.weight, e.LayerNorm.bias, B * L, dt, HID_EPS).view(B, L, -1)
        x = F.dropout(x, self.config.hidden_dropout_prob, self.training)
        km = ops.additive_mask(txt_masks)
        for layer in self.lang_encoder.layer:
            x = layer(x, km)
        return x if self.lang_encoder.update_lang_bert else x.detach()

    def forward_panorama_per_step(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens, pano_masks=None):
        """pano_masks (optional, bool [B, S]): gen_seq_masks(view_lens + obj_lens) made by the caller - static episode buffers hand it over so
        that a captured step contains no length -> mask conversion of its own (duet.buckets)."""
        dt, ie = self.compute_dtype, self.img_embeddings
        B, S, _ = view_img_fts.shape
        if FUSED_EMBED and obj_img_fts is None:        # LN(img linear) + LN(loc linear) + nav-type + token-type -> LN -> dropout: the GEMM + ONE launch
            a = ops.linear(view_img_fts, ie.img_linear.weight, ie.img_linear.bias, out_dtype=dt)
            x = ops.embed_combine(a, dt, ln_a=(ie.img_layer_norm.weight, ie.img_layer_norm.bias),
                                  small=(loc_fts, ie.loc_linear.weight, ie.loc_linear.bias, ie.loc_layer_norm.weight, ie.loc_layer_norm.bias),
                                  row=self.embeddings.token_type_embeddings.weight[1], table=(ie.nav_type_embedding.weight, nav_types.reshape(-1).contiguous()),
                                  ln_o=(ie.layer_norm.weight, ie.layer_norm.bias), eps=HID_EPS, p_drop=self.config.hidden_dropout_prob, training=self.training)
            masks = pano_masks if pano_masks is not None else torch.arange(S, device=x.device)[None, :] < view_lens[:, None]   # gen_seq_masks
            if ie.pano_encoder is not None:
                x = ie.pano_encoder(x, masks)
            return x, masks
        ti = ops.layer_norm(ops.linear(view_img_fts, ie.img_linear.weight, ie.img_linear.bias, out_dtype=dt),
                            ie.img_layer_norm.weight, ie.img_layer_norm.bias, HID_EPS)
        pano_lens = view_lens
        if obj_img_fts is not None:
            # REVERIE / SOON (reference :1096-1114): each sample's objects follow its views. The per-sample slicing + cat +
            # pad_tensors_wgrad loop is ONE row gather: row s of sample b = view s | object s - view_len | zero row.
            lin, ln = (ie.img_linear, ie.img_layer_norm) if ie.obj_linear is None else (ie.obj_linear, ie.obj_layer_norm)
            to = ops.layer_norm(ops.linear(obj_img_fts, lin.weight, lin.bias, out_dtype=dt), ln.weight, ln.bias, HID_EPS)
            V, O, H = S, obj_img_fts.shape[1], ti.shape[-1]
            S = loc_fts.shape[1]                                               # the agent padded loc_fts to max(view+obj)
            dev = ti.device
            rows = torch.cat([torch.zeros(1, H, dtype=ti.dtype, device=dev), ti.reshape(B * V, H), to.reshape(B * O, H)], 0)
            s_idx = torch.arange(S, device=dev)[None, :]
            b_idx = torch.arange(B, device=dev)[:, None]
            vl, ol = view_lens[:, None], obj_lens[:, None]
            idx = torch.where(s_idx < vl, 1 + b_idx * V + s_idx,
                              torch.where(s_idx < vl + ol, 1 + B * V + b_idx * O + (s_idx - vl), torch.zeros_like(s_idx)))
            ti = rows.index_select(0, idx.reshape(-1)).view(B, S, H)
            pano_lens = view_lens + obj_lens
        tl = ops.layer_norm(ops.smallk_linear(loc_fts, ie.loc_linear.weight, ie.loc_linear.bias, dt),
                            ie.loc_layer_norm.weight, ie.loc_layer_norm.bias, HID_EPS)
        srcs = [(ti, "dense", None), (tl, "dense", None),
                (ie.nav_type_embedding.weight, "gather", nav_types.reshape(-1).contiguous()),
                (self.embeddings.token_type_embeddings.weight[1], "bcast", None)]
        x = ops.sum_layer_norm(srcs, ie.layer_norm.weight, ie.layer_norm.bias, B * S, dt, HID_EPS).view(B, S, -1)
        x = ops.dropout(x, self.config.hidden_dropout_prob, self.training)
        masks = pano_masks if pano_masks is not None else torch.arange(S, device=x.device)[None, :] < pano_lens[:, None]   # gen_seq_masks
        if ie.pano_encoder is not None:
            x = ie.pano_encoder(x, masks)
        return x, masks

    def project_text(self, txt_embeds, txt_masks, imagine_embeds=None, imagine_masks=None):
        """The language side of every `navigation` call of an episode, built once by the caller: (per-layer K/V projections of the global
        encoder, of the local encoder, additive key mask) of text (+ imagination tokens). Pass it as batch['text_kv']; an episode tape
        (duet/episode.py:run_episode_taped) does, because a projection made inside a recorded step would belong to step 0 only."""
        c, dt = self.config, self.compute_dtype
        txt, tm = txt_embeds.to(dt), txt_masks
        if c.imagine_enc_pano and c.concat_imagine_with == "language":
            txt, tm = torch.cat([txt, imagine_embeds.to(dt)], 1), torch.cat([tm, imagine_masks], 1)
        txt = txt.contiguous()
        return (self.global_encoder.encoder.project_context(txt), self.local_encoder.encoder.project_context(txt),
                ops.additive_mask(tm).contiguous())

    def forward_navigation_per_step(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks,
                                    gmap_pair_dists, gmap_visited_masks, gmap_vpids, vp_img_embeds, vp_pos_fts, vp_masks,
                                    vp_nav_masks, vp_obj_masks, vp_cand_vpids, imagine_embeds=None, imagine_masks=None, text_kv=None,
                                    fuse_plan=None, masks_add=None):
        """fuse_plan = (src [B,G] int32, bw [B,V] uint8): the caller's own index plan of the global / local fusion (fuse_plan() lists as device
        tensors, e.g. static buffers a captured graph reads); default: built from gmap_vpids / vp_cand_vpids here and cached.
        masks_add = (additive_mask(gmap_masks), additive_mask(vp_masks)) where an episode driver converted the masks of all steps at once."""
        c, dt = self.config, self.compute_dtype
        ge, le = self.global_encoder, self.local_encoder
        B, G = gmap_masks.shape
        if FUSED_EMBED:                 # node image + step embedding + LN(Linear(position)) / view token + LN(Linear(position)): one launch each (:1140-1156)
            gp, vpp = ge.gmap_pos_embeddings, le.vp_pos_embeddings
            gmap = ops.embed_combine(gmap_img_embeds.to(dt), dt, small=(gmap_pos_fts, gp[0].weight, gp[0].bias, gp[1].weight, gp[1].bias),
                                     table=(ge.gmap_step_embeddings.weight, gmap_step_ids.reshape(-1).contiguous()), eps=HID_EPS)
            vp = ops.embed_combine(vp_img_embeds.to(dt), dt, small=(vp_pos_fts, vpp[0].weight, vpp[0].bias, vpp[1].weight, vpp[1].bias), eps=HID_EPS)
        else:
            gmap = gmap_img_embeds.to(dt) + ge.gmap_step_embeddings.weight.index_select(0, gmap_step_ids.reshape(-1)).view(*gmap_step_ids.shape, -1).to(dt) \
                + ge.gmap_pos_embeddings.embed(gmap_pos_fts, dt)
            vp = vp_img_embeds.to(dt) + le.vp_pos_embeddings.embed(vp_pos_fts, dt)
        sprels = None
        if ge.sprel_linear is not None:                                                     # reference :1145-1147
            sprels = torch.addcmul(ge.sprel_linear.bias[0], gmap_pair_dists, ge.sprel_linear.weight[0, 0])
        def language_side():
            """text (+ imagination tokens) and its additive key mask (reference :1110-1125)"""
            txt, tm = txt_embeds.to(dt), txt_masks
            if c.imagine_enc_pano and c.concat_imagine_with == "language":
                assert imagine_embeds is not None and imagine_masks is not None
                txt, tm = torch.cat([txt, imagine_embeds.to(dt)], 1), torch.cat([tm, imagine_masks], 1)
            else:
                assert not c.imagine_enc_pano
            return txt.contiguous(), ops.additive_mask(tm).contiguous()

        kv_g = kv_l = None
        if text_kv is not None:
            kv_g, kv_l, lm = text_kv
            txt = None
        elif CACHE_TEXT_KV:
            # the language stream is never updated (use_lang2visn_attn False), so its per-layer K/V projections are the same
            # for every step of an episode: project once, reduce their gradient once - and build the concatenated stream and its
            # mask only then. Key = identity of the caller-held embeddings and masks (strong refs keep the ids unique) + the
            # parameter epoch; a backward through the entry drops it.
            wk = ge.encoder.x_layers[0].visual_attention.att.key.weight
            key = (id(txt_embeds), id(imagine_embeds), id(txt_masks), id(imagine_masks), txt_embeds._version, txt_masks._version,
                   imagine_masks._version if torch.is_tensor(imagine_masks) else None, torch.is_grad_enabled(), dt, ops.SHADOWS.epoch,
                   wk._version)
            ent = self._kv_cache
            if ent is None or ent[0] != key:
                txt, lm = language_side()
                kvs = (ge.encoder.project_context(txt), le.encoder.project_context(txt))
                ent = self._kv_cache = (key, (txt_embeds, imagine_embeds, txt_masks, imagine_masks), kvs[0], kvs[1], txt, lm)
                for kv in kvs[0] + kvs[1]:
                    if kv.requires_grad:
                        kv.register_hook(self._drop_kv_cache)
            kv_g, kv_l, txt, lm = ent[2], ent[3], ent[4], ent[5]
        else:
            txt, lm = language_side()
        gmap, vp = gmap.contiguous(), vp.contiguous()
        gm, vm = masks_add if masks_add is not None else (ops.additive_mask(gmap_masks), ops.additive_mask(vp_masks))
        if DUAL_BRANCHES and kv_g is not None:
            # the global-map and local-viewpoint branches are independent and have the same layer shapes: layer i of both runs as
            # dual-problem GEMM launches (each branch alone is 2-10 row tiles, far below one wave of CUs)
            for i, (lg, ll) in enumerate(zip(ge.encoder.x_layers, le.encoder.x_layers)):
                gmap, vp = ops.dual_xatt_q_block(gmap, vp, kv_g[i], kv_l[i], lm, _att(lg.visual_attention), _att(ll.visual_attention),
                                                 drop0=_drop(lg.visual_attention), drop1=_drop(ll.visual_attention))
                gmap, vp = ops.dual_self_att_block(gmap, vp, gm, vm, _att(lg.visn_self_att), _att(ll.visn_self_att),
                                                   drop0=_drop(lg.visn_self_att), drop1=_drop(ll.visn_self_att), bias0=sprels)
                gmap, vp = ops.dual_ffn_block(gmap, vp, _ffn(lg.visn_inter, lg.visn_output), _ffn(ll.visn_inter, ll.visn_output),
                                              drop0=_drop(lg.visn_output), drop1=_drop(ll.visn_output))
        else:
            gmap = ge.encoder(txt, lm, gmap, gm, sprels, kvs=kv_g)
            vp = le.encoder(txt, lm, vp, vm, kvs=kv_l)
        # reference :1185-1217: fuse weight, the two masked heads and the global / local fusion - one launch (ops.duet_heads)
        f = None if self.sap_fuse_linear is None else self.sap_fuse_linear(torch.cat([gmap[:, 0], vp[:, 0]], 1).contiguous())
        src, bw = fuse_plan if fuse_plan is not None else self._fuse_plan(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, vp.shape[1])
        global_logits, local_logits, fused_logits = ops.duet_heads(self.global_sap_head(gmap), self.local_sap_head(vp), f,
                                                                   gmap_visited_masks, gmap_masks, vp_nav_masks, src, bw)
        obj_logits = self.og_head(vp, ~vp_obj_masks) if vp_obj_masks is not None else None      # reference :1220-1225
        return {"gmap_embeds": gmap, "vp_embeds": vp, "global_logits": global_logits, "local_logits": local_logits,
                "fused_logits": fused_logits, "obj_logits": obj_logits}

    @staticmethod
    def fuse_plan(gmap_vpids, visited, vp_cand_vpids, G, V):
        """Host half of the fusion: src[i][g] = local candidate index that IS unvisited map node g, -2 = unvisited node no
        candidate points at (takes the summed logits of the visited candidates), -1 = nothing; bw[i][j] = candidate j is visited."""
        B = len(gmap_vpids)
        src = [[-1] * G for _ in range(B)]
        bw = [[0] * V for _ in range(B)]
        for i in range(B):
            seen = {vp for vp, m in zip(gmap_vpids[i], visited[i]) if m}
            cand = {}
            for j, cv in enumerate(vp_cand_vpids[i]):
                if j > 0:
                    if cv in seen:
                        bw[i][j] = 1
                    else:
                        cand[cv] = j
            for j, vp in enumerate(gmap_vpids[i]):
                if j > 0 and vp not in seen:
                    src[i][j] = cand[vp] if vp in cand else -2
        return src, bw

    def _fuse_plan(self, gmap_vpids, visited_masks, vp_cand_vpids, G, V):
        """Index plan of the fusion (reference :1198-1217: fused[i,0] = g+l (stop); an unvisited map node takes the local logit of the
        candidate that IS that node, otherwise the summed local logits of the already-visited candidates (backtrack)) as device tensors,
        cached per (vpid lists, visited mask) identity, so a teacher-forced / replayed episode pays the device->host read of the visited
        mask once. The arithmetic is part of ops.duet_heads."""
        B = len(gmap_vpids)
        key = (id(gmap_vpids), id(vp_cand_vpids), id(visited_masks), visited_masks._version, B, G, V)
        hit = self._fuse_plans.get(key)
        if hit is None:
            src, bw = self.fuse_plan(gmap_vpids, visited_masks.tolist(), vp_cand_vpids, G, V)
            if len(self._fuse_plans) >= 64:
                self._fuse_plans.clear()
            dev = visited_masks.device
            hit = self._fuse_plans[key] = (torch.tensor(src, dtype=torch.int32, device=dev), torch.tensor(bw, dtype=torch.uint8, device=dev),
                                           (gmap_vpids, vp_cand_vpids, visited_masks))     # strong refs keep the ids unique
        return hit[0], hit[1]

    def _fuse(self, gl, ll, gmap_vpids, visited_masks, vp_cand_vpids):
        """Fusion of already-masked logits alone (kept for callers that hold the two logit tensors)."""
        src, bw = self._fuse_plan(gmap_vpids, visited_masks, vp_cand_vpids, gl.shape[1], ll.shape[1])
        return ops.duet_fuse(gl, ll, src, bw)

    def forward(self, mode, batch, **kwargs):
        c = self.config
        if mode == "language":
            return self.forward_text(batch["txt_ids"], batch["txt_masks"])
        if mode == "imagine":
            assert c.imagine_enc_pano and batch["imagine_feats"] is not None
            return self.imagine_embeddings(batch["imagine_feats"], batch["imagine_masks"], self.compute_dtype)
        if mode == "align_with_contrastive_loss":
            assert c.imagine_enc_pano
            txt = batch["align_txt_embeds"]
            return self.contrastive_alignment_model(
                align_txt_embeds=txt.detach() if c.fix_lang_inside_cosine_model else txt, txt_masks=batch["txt_masks"],
                align_imagine_embeds=batch["align_imagine_embeds"], imagine_masks=batch["imagine_masks"],
                sub_instr_segs=batch.get("sub_instr_segs"), sub_instr_imag_flag=batch.get("sub_instr_imag_flag"),
                noun_phrase_segs=batch.get("noun_phrase_segs"), obs_instr_ids=batch.get("obs_instr_ids"))
        if mode == "panorama":
            return self.forward_panorama_per_step(batch["view_img_fts"], batch.get("obj_img_fts"), batch["loc_fts"],
                                                  batch["nav_types"], batch["view_lens"], batch.get("obj_lens"), batch.get("pano_masks"))
        if mode == "navigation":
            return self.forward_navigation_per_step(
                batch["txt_embeds"], batch["txt_masks"], batch["gmap_img_embeds"], batch["gmap_step_ids"],
                batch["gmap_pos_fts"], batch["gmap_masks"], batch["gmap_pair_dists"], batch["gmap_visited_masks"],
                batch["gmap_vpids"], batch["vp_img_embeds"], batch["vp_pos_fts"], batch["vp_masks"], batch["vp_nav_masks"],
                batch.get("vp_obj_masks"), batch["vp_cand_vpids"], imagine_embeds=batch.get("imagine_embeds"),
                imagine_masks=batch.get("imagine_masks"), text_kv=batch.get("text_kv"), fuse_plan=batch.get("fuse_plan"),
                masks_add=batch.get("masks_add"))
        raise NotImplementedError("wrong mode: %s" % mode)
