"""Pre-norm panorama encoder holders (drop-in for the used part of VLN-DUET/map_nav_src/models/transformer.py:
TransformerEncoder :61-89 + TransformerEncoderLayer.forward_pre :170-182; the decoder classes of the DETR copy are
never instantiated by DUET and are not provided). Parameter names match nn.MultiheadAttention's packed layout
(self_attn.in_proj_weight [2304,768], in_proj_bias, out_proj.*) so checkpoints load by key."""
import torch
from torch import nn

from vln_imagine_amd import ops


class _PackedSelfAttn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * h, h))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * h))
        self.out_proj = nn.Linear(h, h)
        nn.init.xavier_uniform_(self.in_proj_weight)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward, dropout=0.1):
        super().__init__()
        self.p = dropout
        self.self_attn = _PackedSelfAttn(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)          # eps 1e-5 (nn.LayerNorm default), unlike BertLayerNorm
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x, key_add_mask):
        a = self.self_attn
        x = ops.prenorm_att_block(x, key_add_mask, (self.norm1.weight, self.norm1.bias, a.in_proj_weight, a.in_proj_bias,
                                                    a.out_proj.weight, a.out_proj.bias), eps=self.norm1.eps,
                                  drop=ops.drop_cfg(self.p, self.p, self.training))
        return ops.prenorm_ffn_block(x, (self.norm2.weight, self.norm2.bias, self.linear1.weight, self.linear1.bias,
                                         self.linear2.weight, self.linear2.bias), eps=self.norm2.eps,
                                     drop=ops.drop_cfg(0.0, self.p, self.training))


class TransformerEncoder(nn.Module):
    def __init__(self, config, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderLayer(config.hidden_size, config.intermediate_size,
                                                             config.hidden_dropout_prob)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(config.hidden_size, eps=1e-12)

    def forward(self, x, valid_mask):
        """valid_mask [B,S] bool (True = real view). Padded keys get -inf (src_key_padding_mask semantics)."""
        km = ops.additive_mask(valid_mask, inf=True)
        for l in self.layers:
            x = l(x, km)
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
