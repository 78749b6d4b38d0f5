"""Builds DUET's GlocalTextPathNavCMT from run arguments (drop-in for
VLN-DUET/map_nav_src/models/vlnbert_init.py:4-77), config built offline."""
import torch


def get_tokenizer(args):
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained("xlm-roberta-base" if getattr(args, "tokenizer", None) == "xlm" else "bert-base-uncased")


def config_from_args(args):
    from vln_imagine_amd.duet.config import DuetConfig
    g = lambda k, d=None: getattr(args, k, d)
    kw = dict(max_action_steps=100, image_feat_size=g("image_feat_size", 768), angle_feat_size=g("angle_feat_size", 4),
              obj_feat_size=g("obj_feat_size", 0), obj_loc_size=3, num_l_layers=g("num_l_layers", 9),
              num_pano_layers=g("num_pano_layers", 2), num_x_layers=g("num_x_layers", 4), graph_sprels=g("graph_sprels", False),
              glocal_fuse=g("fusion") == "dynamic", fix_lang_embedding=g("fix_lang_embedding", False),
              fix_pano_embedding=g("fix_pano_embedding", False), fix_local_branch=g("fix_local_branch", False),
              update_lang_bert=not g("fix_lang_embedding", False), output_attentions=True, pred_head_dropout_prob=0.1,
              use_lang2visn_attn=False, imagine_enc_pano=g("imagine_enc_pano", False))
    if kw["imagine_enc_pano"]:
        kw.update(max_imagination_len=g("max_imagination_len", 20), fix_imagine_embeds=g("fix_imagine_embeds", False),
                  bypass_imag_encoder=g("bypass_imag_encoder", False), use_cosine_aux_loss=g("use_cosine_aux_loss", False),
                  concat_imagine_with=g("concat_imagine_with", "language"),
                  fix_lang_inside_cosine_model=g("fix_lang_inside_cosine_model", False), aux_loss_type=g("aux_loss_type", "cosine"),
                  infonce_temperature=g("infonce_temperature", 0.3), no_loss_test=g("no_loss_test", False),
                  dataset=g("dataset", "r2r"))
    else:
        kw.update(use_cosine_aux_loss=False, no_loss_test=False)
    if g("tokenizer") == "xlm":
        kw.update(vocab_size=250002, type_vocab_size=2, max_position_embeddings=514, pad_token_id=1)
    return DuetConfig(**kw)


def get_vlnbert_models(args, config=None):
    from .vilmodel import GlocalTextPathNavCMT
    weights = {}
    path = getattr(args, "bert_ckpt_file", None)
    if path is not None:
        for k, v in torch.load(path, map_location="cpu").items():
            k = k[7:] if k.startswith("module") else k
            weights["bert." + k if ("_head" in k or "sap_fuse" in k) else k] = v
    return GlocalTextPathNavCMT.from_pretrained(None, config=config or config_from_args(args), state_dict=weights)
