"""state_dict key -> shape table of the HAMT NavCMT, in the reference's registration order
(VLN-HAMT/finetune_src/models/vilmodel_cmt.py:967-989). Key names are ABI: released
checkpoints are loaded by key (r2r/agent_cmt.py:854-875)."""
from collections import OrderedDict


def _lin(d, p, o, i, bias=True):
    d[p + ".weight"] = (o, i)
    if bias:
        d[p + ".bias"] = (o,)


def _ln(d, p, h):
    d[p + ".weight"] = (h,)
    d[p + ".bias"] = (h,)


def _bert_attention(d, p, h, att="self"):
    for n in ("query", "key", "value"):
        _lin(d, f"{p}.{att}.{n}", h, h)
    _lin(d, p + ".output.dense", h, h)
    _ln(d, p + ".output.LayerNorm", h)


def _bert_layer(d, p, h, ff):
    _bert_attention(d, p + ".attention", h)
    _lin(d, p + ".intermediate.dense", ff, h)
    _lin(d, p + ".output.dense", h, ff)
    _ln(d, p + ".output.LayerNorm", h)


def param_shapes(cfg):
    h, ff = cfg.hidden_size, cfg.intermediate_size
    d = OrderedDict()
    d["embeddings.word_embeddings.weight"] = (cfg.vocab_size, h)
    d["embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, h)
    d["embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, h)
    _ln(d, "embeddings.LayerNorm", h)
    p = "img_embeddings"
    _lin(d, p + ".img_linear", h, cfg.image_feat_size); _ln(d, p + ".img_layer_norm", h)
    _lin(d, p + ".ang_linear", h, cfg.angle_feat_size); _ln(d, p + ".ang_layer_norm", h)
    d[p + ".nav_type_embedding.weight"] = (3, h)
    _ln(d, p + ".layer_norm", h)
    p = "hist_embeddings"
    d[p + ".cls_token"] = (1, 1, h)
    _lin(d, p + ".img_linear", h, cfg.image_feat_size); _ln(d, p + ".img_layer_norm", h)
    _lin(d, p + ".ang_linear", h, cfg.angle_feat_size); _ln(d, p + ".ang_layer_norm", h)
    d[p + ".position_embeddings.weight"] = (cfg.max_action_steps, h)
    d[p + ".type_embedding.weight"] = (1, h)
    _ln(d, p + ".layer_norm", h)
    if cfg.hist_enc_pano:
        _lin(d, p + ".pano_img_linear", h, cfg.image_feat_size); _ln(d, p + ".pano_img_layer_norm", h)
        _lin(d, p + ".pano_ang_linear", h, cfg.angle_feat_size); _ln(d, p + ".pano_ang_layer_norm", h)
        for i in range(cfg.num_h_pano_layers):
            _bert_layer(d, f"{p}.pano_encoder.layer.{i}", h, ff)
    if cfg.imagine_enc_pano and (cfg.use_cosine_aux_loss or cfg.no_loss_test):
        q = "contrastive_alignment_model.image_proj"
        _lin(d, q + ".fc1", 512, 768, bias=False)
        _lin(d, q + ".fc2", 512, 512, bias=False)
        _lin(d, q + ".fc3", h, 512, bias=False)
    if cfg.imagine_enc_pano:
        p = "imagine_embeddings"
        if cfg.bypass_imag_encoder:
            d[p + ".type_embedding.weight"] = (1, h)
        else:
            d[p + ".position_embeddings.weight"] = (cfg.max_imagination_len, h)
            d[p + ".type_embedding.weight"] = (1, h)
            _ln(d, p + ".layer_norm", h)
            _lin(d, p + ".pano_img_linear", h, cfg.image_feat_size); _ln(d, p + ".pano_img_layer_norm", h)
            for i in range(cfg.num_h_pano_layers):
                _bert_layer(d, f"{p}.pano_encoder.layer.{i}", h, ff)
    for i in range(cfg.num_l_layers):
        _bert_layer(d, f"encoder.layer.{i}", h, ff)
    for i in range(cfg.num_h_layers):               # registration order of LxmertEncoder (:458-473): layer, h_layers, r_layers, x_layers
        _bert_layer(d, f"encoder.h_layers.{i}", h, ff)
    for i in range(cfg.num_r_layers):
        _bert_layer(d, f"encoder.r_layers.{i}", h, ff)
    for i in range(cfg.num_x_layers):
        p = f"encoder.x_layers.{i}"
        for side in ("lang", "visn"):
            _bert_attention(d, f"{p}.{side}_self_att", h)
            _lin(d, f"{p}.{side}_inter.dense", ff, h)
            _lin(d, f"{p}.{side}_output.dense", h, ff)
            _ln(d, f"{p}.{side}_output.LayerNorm", h)
        _bert_attention(d, f"{p}.visual_attention", h, att="att")
    _lin(d, "next_action.net.0", h, h)
    _ln(d, "next_action.net.2", h)
    _lin(d, "next_action.net.4", 1, h)
    return d
