"""Builds the HAMT NavCMT from run arguments (drop-in for
VLN-HAMT/finetune_src/models/vlnbert_init.py:4-83) without touching the network:
bert-base-uncased's config values are constants (vln_imagine_amd/hamt/config.py)."""
import torch


def get_tokenizer(args):
    from transformers import AutoTokenizer
    name = "xlm-roberta-base" if (getattr(args, "dataset", None) == "rxr" or getattr(args, "tokenizer", None) == "xlm") \
        else "bert-base-uncased"
    return AutoTokenizer.from_pretrained(name)


def config_from_args(args):
    from vln_imagine_amd.hamt.config import HamtConfig
    g = lambda k, d=None: getattr(args, k, d)
    kw = dict(
        image_feat_size=g("image_feat_size", 768), angle_feat_size=g("angle_feat_size", 4),
        num_l_layers=g("num_l_layers", 9), num_r_layers=0, num_h_layers=g("num_h_layers", 0),
        num_x_layers=g("num_x_layers", 4), hist_enc_pano=g("hist_enc_pano", False),
        num_h_pano_layers=g("hist_pano_num_layers", 2),
        fix_lang_embedding=g("fix_lang_embedding", False), fix_hist_embedding=g("fix_hist_embedding", False),
        fix_obs_embedding=g("fix_obs_embedding", False), update_lang_bert=not g("fix_lang_embedding", False),
        output_attentions=True, pred_head_dropout_prob=0.1, no_lang_ca=g("no_lang_ca", False),
        act_pred_token=g("act_pred_token", "ob_txt"), max_action_steps=50,
        imagine_enc_pano=g("imagine_enc_pano", False))
    if kw["imagine_enc_pano"]:
        kw.update(max_imagination_len=g("max_imagination_len", 20), fix_imagine_embeds=g("fix_imagine_embeds", False),
                  bypass_imag_encoder=g("bypass_imag_encoder", False), use_cosine_aux_loss=g("use_cosine_aux_loss", False),
                  aux_loss_type=g("aux_loss_type", "cosine"), infonce_temperature=g("infonce_temperature", 0.3),
                  contrastive_margin_value=g("contrastive_margin_value", 1.0),
                  concat_imagine_with=g("concat_imagine_with", "language"), no_loss_test=g("no_loss_test", False))
    else:
        kw.update(use_cosine_aux_loss=False, no_loss_test=False)
    if g("dataset") == "rxr" or g("tokenizer") == "xlm":
        kw.update(vocab_size=250002, type_vocab_size=2, max_position_embeddings=514, pad_token_id=1)
    return HamtConfig(**kw)


def get_vlnbert_models(args, config=None):
    from .vilmodel_cmt import NavCMT
    weights = {}
    path = getattr(args, "bert_ckpt_file", None)
    if path is not None:
        for k, v in torch.load(path, map_location="cpu").items():
            if k.startswith("module"):
                weights[k[7:]] = v
            else:
                weights["bert." + k if k.startswith("next_action") else k] = v
    return NavCMT.from_pretrained(None, config=config or config_from_args(args), state_dict=weights)
