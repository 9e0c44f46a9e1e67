"""Layer mirrors of models/simple_siamese/layers.py (class names, constructor arguments, parameter names and
initialisation), running on the HIP kernels of csrc/review_bag.hip, csrc/pair_head.hip and csrc/dense_misc.hip."""
import torch
import torch.nn as nn

from ... import functional as RF
from ..deepconn.layers import FM as _FM
from ..deepconn.layers import LastFeat as _LastFeat
from ..deepconn.layers import WordEmbedding as _WordEmbedding


class NodeDropout(torch.nn.Dropout):
    """layers.py:7-22 -- one keep/drop decision per (batch row, node): [bz, seq_len, dim] * mask[bz, seq_len, 1]."""

    def multiplier(self, bz, seq_len, device):
        return RF.dropout_multiplier((bz, seq_len), self.p, self.training, device)

    def forward(self, input_tensor):
        m = self.multiplier(input_tensor.shape[0], input_tensor.shape[1], input_tensor.device)
        return input_tensor if m is None else m.unsqueeze(2) * input_tensor


class VariationalDropout(torch.nn.Dropout):
    """layers.py:24-50 -- one mask per (batch row, embedding dim), shared by every time step."""

    def multiplier(self, bz, dim, device):
        return RF.dropout_multiplier((bz, dim), self.p, self.training, device)

    def forward(self, input_tensor):
        m = self.multiplier(input_tensor.shape[0], input_tensor.shape[-1], input_tensor.device)
        return input_tensor if m is None else m.unsqueeze(1) * input_tensor


class WordEmbedding(_WordEmbedding):
    """layers.py:53-68 (adds the `sparse` flag; sparse gradients are not produced here: the HIP backward is dense)."""

    def __init__(self, vocab_size, embedding_dim, pretrained_embeddings=None, padding_idx=0, freeze_embeddings=False,
                 sparse=False):
        super().__init__(vocab_size, embedding_dim, pretrained_embeddings=pretrained_embeddings, padding_idx=padding_idx,
                         freeze_embeddings=freeze_embeddings)
        if sparse:
            raise ValueError("sparse embedding gradients are not supported by the HIP path")


class MaskedAvgPooling1d(nn.Module):
    """layers.py:90-110 -- inputs [bz, hdim, seq_len], masks [bz, seq_len] -> [bz, hdim, 1] = masked sum / (count + 1e-8).
    Standalone form on materialised rows; the model fuses the lookup and this pooling (RF.review_bag)."""

    def forward(self, inputs, input_masks):
        assert input_masks.dim() == 2
        bz, hdim, seq_len = inputs.shape
        # a [bz*seq_len, hdim] table indexed by position is exactly the bag kernel's input
        table = inputs.transpose(1, 2).reshape(bz * seq_len, hdim)
        ids = torch.arange(bz * seq_len, device=inputs.device, dtype=torch.int64).view(bz, seq_len)
        return RF.review_bag(table, ids, input_masks, padding_idx=None).unsqueeze(2)


class AddictiveAttention(nn.Module):
    """layers.py:171-197 -- tanh projection, inner product, masked softmax over the sequence, weighted sum."""

    def __init__(self, hidden_dim, latent_dim):
        super().__init__()
        self.proj_layer = nn.Sequential(nn.Linear(hidden_dim, latent_dim), nn.Tanh())
        self.inner_product = nn.Linear(latent_dim, 1, bias=False)

    def forward(self, inputs, input_masks, node_drop=None):
        """inputs [bz, seq_len, hdim], input_masks [bz, seq_len] -> (outputs [bz, hdim], att_scores [bz, seq_len, 1])."""
        assert input_masks.dim() == 2
        lin = self.proj_layer[0]
        return RF.additive_attention(inputs, input_masks, lin.weight, lin.bias, self.inner_product.weight, node_drop=node_drop)


class LastFeat(_LastFeat):
    """layers.py:234-261 -- W [feat, latent], b [latent] (starts at 0 here), ebd [vocab, latent]; out = feat @ W + b + ebd[id]."""
    BIAS_INIT = 0.0


class FM(_FM):
    """layers.py:299-343 -- h [latent,1], user_bias [U,1], item_bias [I,1], g_bias [1] (starts at 4.0 in this model)."""
    G_BIAS_INIT = 4.0


class FMWithoutUIBias(_FM):
    """layers.py:263-297 -- h [latent,1], g_bias [1] (4.0); pred = dropout(relu(u*i)) @ h + g_bias, no user / item biases."""
    G_BIAS_INIT = 4.0
    WITH_ID_BIASES = False

    def __init__(self, user_size, item_size, latent_dim, dropout, user_padding_idx, item_padding_idx):
        super().__init__(user_size, item_size, latent_dim, dropout, user_padding_idx, item_padding_idx)
        # the head kernel always adds bias rows: frozen zero tables, outside the state_dict
        self.register_buffer("_zero_user_bias", torch.zeros(user_size, 1), persistent=False)
        self.register_buffer("_zero_item_bias", torch.zeros(item_size, 1), persistent=False)


def rating_head(user_feat: LastFeat, item_feat: LastFeat, fm, u_text_feat, i_text_feat, u_ids, i_ids):
    """LastFeat(user) + LastFeat(item) + FM / FMWithoutUIBias in one HIP kernel pair.
    i_text_feat None: u_text_feat holds both towers, [2*bz, F] with the user rows first."""
    if fm.training and fm.dropout.p < 1.0 and torch.is_grad_enabled():
        drop = float(fm.dropout.p)       # training forward: the head kernel draws the mask (and clears its gradient buffer) itself
    else:
        drop = RF.dropout_multiplier((u_ids.shape[0], fm.h.shape[0]), fm.dropout.p, fm.training, u_text_feat.device)
    ub = fm.user_bias.weight if fm.WITH_ID_BIASES else fm._zero_user_bias
    ib = fm.item_bias.weight if fm.WITH_ID_BIASES else fm._zero_item_bias
    return RF.pair_head(u_text_feat, i_text_feat, u_ids, i_ids,
                        user_feat.W, user_feat.b, user_feat.ebd.weight,
                        item_feat.W, item_feat.b, item_feat.ebd.weight,
                        fm.h, fm.g_bias, ub, ib, drop=drop, pad_u=fm.user_padding_idx, pad_i=fm.item_padding_idx)
