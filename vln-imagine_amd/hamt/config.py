"""HAMT model configuration without the network.

The reference builds a HuggingFace `PretrainedConfig.from_pretrained('bert-base-uncased')`
and copies run flags onto it (VLN-HAMT/finetune_src/models/vlnbert_init.py:37-76).
bert-base-uncased's values are constants, restated here so no download is needed.
"""

BERT_BASE = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
    intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
    attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
    layer_norm_eps=1e-12, initializer_range=0.02, pad_token_id=0,
)


def hamt_config_dict(**over):
    """Attributes vlnbert_init.py:43-76 sets, defaulting to the shipped run
    (scripts/run_r2r.sh:18-76) except the freeze flags, which default to trainable."""
    d = dict(
        image_feat_size=768, angle_feat_size=4,
        num_l_layers=9, num_r_layers=0, num_h_layers=0, num_x_layers=4,
        hist_enc_pano=True, num_h_pano_layers=2,
        fix_lang_embedding=False, fix_hist_embedding=False, fix_obs_embedding=False,
        update_lang_bert=True, output_attentions=True, output_hidden_states=False,
        pred_head_dropout_prob=0.1, no_lang_ca=False, act_pred_token="ob_txt",
        max_action_steps=50,
        imagine_enc_pano=True, max_imagination_len=20, fix_imagine_embeds=False,
        bypass_imag_encoder=True, use_cosine_aux_loss=True, aux_loss_type="cosine",
        infonce_temperature=0.3, contrastive_margin_value=1.0,
        concat_imagine_with="language", no_loss_test=False,
    )
    for k in over:
        if k not in d and k not in BERT_BASE:
            raise KeyError(f"unknown HAMT config key {k!r}")
    d.update(over)
    return d


class HamtConfig:
    """Plain attribute bag: BERT-base constants + HAMT flags."""

    def __init__(self, **over):
        self.__dict__.update(BERT_BASE)
        self.__dict__.update(hamt_config_dict(**{k: v for k, v in over.items() if k not in BERT_BASE}))
        self.__dict__.update({k: v for k, v in over.items() if k in BERT_BASE})

    def to_dict(self):
        return dict(self.__dict__)
