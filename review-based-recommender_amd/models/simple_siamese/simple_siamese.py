"""SimpleSiamese with the reference's constructor / forward signature and state_dict keys
(models/simple_siamese/simple_siamese.py:8-87), running on the HIP kernels of csrc/ (SURVEY.md 8 f-4)."""
import torch
import torch.nn as nn

from ... import functional as RF
from .layers import (FM, AddictiveAttention, FMWithoutUIBias, LastFeat, MaskedAvgPooling1d, NodeDropout, VariationalDropout,
                     WordEmbedding, rating_head)
from .utils import get_rev_mask  # noqa: F401  (re-exported as in the reference module)


class SimpleSiamese(nn.Module):
    def __init__(self, embedding_dim, latent_dim, vocab_size, user_size, item_size, pretrained_embeddings, freeze_embeddings,
                 dropout, word_dropout, review_dropout, use_ui_bias, latent_transform):
        super().__init__()
        self.use_ui_bias = use_ui_bias
        self.embedding_dim = embedding_dim
        self.latent_transform = latent_transform
        self.vocab_size, self.user_size, self.item_size = vocab_size, user_size, item_size
        self.validate_ids = True      # device-side range check of every id tensor (functional.sanitize_ids), see DeepCoNNpp

        self.word_embedding = WordEmbedding(vocab_size, embedding_dim, pretrained_embeddings=pretrained_embeddings,
                                            freeze_embeddings=freeze_embeddings, padding_idx=0)
        self.var_dropout = VariationalDropout(p=word_dropout)
        self.review_dropout = NodeDropout(p=review_dropout)
        self.masked_pooling_1d = MaskedAvgPooling1d()
        if self.latent_transform:
            self.latent_transform_layer = nn.Sequential(nn.Linear(embedding_dim, latent_dim), nn.Tanh())
        feat = latent_dim if self.latent_transform else embedding_dim
        self.user_last_feat_layer = LastFeat(user_size, feat, latent_dim, padding_idx=0)
        self.item_last_feat_layer = LastFeat(item_size, feat, latent_dim, padding_idx=0)
        self.review_att_layer = AddictiveAttention(feat, latent_dim)
        fm_cls = FM if self.use_ui_bias else FMWithoutUIBias
        self.fm = fm_cls(user_size, item_size, latent_dim, dropout, user_padding_idx=0, item_padding_idx=0)

    def _tower(self, revs, word_masks, rev_masks):
        bz, rv_num, rv_len = revs.shape
        dev = revs.device
        # word lookup + variational dropout + masked average in one pass over the table rows
        drop = self.var_dropout.multiplier(bz * rv_num, self.embedding_dim, dev)
        rev = RF.review_bag(self.word_embedding.weight, revs.reshape(bz * rv_num, rv_len),
                            word_masks.reshape(bz * rv_num, rv_len), drop=drop, padding_idx=self.word_embedding.padding_idx)
        if self.latent_transform:
            lin = self.latent_transform_layer[0]
            rev = RF.linear(rev, lin.weight, lin.bias, tanh=True)
        rev = rev.view(bz, rv_num, -1)
        node = self.review_dropout.multiplier(bz, rv_num, dev)
        feat, _ = self.review_att_layer(rev, rev_masks, node_drop=node)
        return feat

    def forward(self, u_revs, i_revs, u_rev_word_masks, i_rev_word_masks, u_rev_masks, i_rev_masks, u_ids, i_ids):
        """u_revs / i_revs [bz, rv_num, rv_len] int64, word masks [bz, rv_num, rv_len], review masks [bz, rv_num], ids [bz]
        -> (out_logits [bz], None, None), as the reference returns."""
        bz = u_revs.shape[0]
        stacked = None
        if self.validate_ids:
            same = u_revs.shape == i_revs.shape
            outs = RF.sanitize_ids([(u_revs, self.vocab_size, 0), (i_revs, self.vocab_size, 0), (u_ids, self.user_size, 0),
                                    (i_ids, self.item_size, 0)], stack_first_two=same)
            if same:
                stacked, u_ids, i_ids = outs
                u_revs, i_revs = stacked[:bz], stacked[bz:]
            else:
                u_revs, i_revs, u_ids, i_ids = outs
        if u_revs.shape == i_revs.shape and u_rev_word_masks.shape == i_rev_word_masks.shape and u_rev_masks.shape == i_rev_masks.shape:
            # the towers share every layer (simple_siamese.py:58-77: one word_embedding, latent_transform_layer, review_att_layer): both sides go through each kernel as ONE batch of
            # 2*bz rows, user rows first, and the head takes the stacked features whole
            u_rev_feat = self._tower(stacked if stacked is not None else RF.stack_rows(u_revs, i_revs),
                                     RF.stack_rows(u_rev_word_masks, i_rev_word_masks),
                                     RF.stack_rows(u_rev_masks, i_rev_masks))
            i_rev_feat = None
        else:
            u_rev_feat = self._tower(u_revs, u_rev_word_masks, u_rev_masks)
            i_rev_feat = self._tower(i_revs, i_rev_word_masks, i_rev_masks)
        out_logits = rating_head(self.user_last_feat_layer, self.item_last_feat_layer, self.fm, u_rev_feat, i_rev_feat,
                                 u_ids, i_ids)
        return out_logits.view(bz), None, None
