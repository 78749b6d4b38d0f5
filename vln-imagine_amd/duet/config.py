"""DUET model configuration without the network (VLN-DUET/map_nav_src/models/vlnbert_init.py:32-70)."""
from vln_imagine_amd.hamt.config import BERT_BASE


def duet_config_dict(**over):
    """Attributes vlnbert_init.py:39-70 sets; defaults = the shipped run (scripts/run_r2r.sh:28-81)."""
    d = dict(
        max_action_steps=100, image_feat_size=768, angle_feat_size=4, obj_feat_size=0, obj_loc_size=3,
        num_l_layers=9, num_pano_layers=2, num_x_layers=4, graph_sprels=True, glocal_fuse=True,
        fix_lang_embedding=False, fix_pano_embedding=False, fix_local_branch=False, update_lang_bert=True,
        output_attentions=True, output_hidden_states=False, pred_head_dropout_prob=0.1, use_lang2visn_attn=False,
        imagine_enc_pano=True, max_imagination_len=20, fix_imagine_embeds=False, bypass_imag_encoder=True,
        use_cosine_aux_loss=True, concat_imagine_with="language", fix_lang_inside_cosine_model=True,
        aux_loss_type="cosine", infonce_temperature=0.3, contrastive_margin_value=1.0, no_loss_test=False,
        dataset="r2r",
    )
    for k in over:
        if k not in d and k not in BERT_BASE:
            raise KeyError(f"unknown DUET config key {k!r}")
    d.update(over)
    return d


class DuetConfig:
    def __init__(self, **over):
        self.__dict__.update(BERT_BASE)
        self.__dict__.update(duet_config_dict(**{k: v for k, v in over.items() if k not in BERT_BASE}))
        self.__dict__.update({k: v for k, v in over.items() if k in BERT_BASE})

    def to_dict(self):
        return dict(self.__dict__)
