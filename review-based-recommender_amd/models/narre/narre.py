"""NARRE with the reference's constructor / forward signature and state_dict keys
(models/narre/narre.py:139-192), running on the HIP kernels of csrc/.

narre.py defines its OWN WordEmbedding / LinearAttention / LastFeat / FM (narre.py:9-137), with
b = g_bias = 0.1 initialisation; those are mirrored here (the LastFeat / FM mirrors of
models/deepconn/layers.py initialise identically, so they are shared)."""
import torch
import torch.nn as nn

from ... import functional as RF
from ..deepconn.layers import FM, LastFeat, WordEmbedding, rating_head
from .layers import NgramFeat


class LinearAttention(nn.Module):
    """narre.py:26-64: review-level attention conditioned on the counterpart-id embedding.
    att = exp(logit) / (sum_R exp(logit) + 1e-8): un-masked, no max subtraction (quirk 3)."""

    def __init__(self, vocab_size, feat_size, hidden_dim, dropout, padding_idx=0):
        super().__init__()
        self.padding_idx = padding_idx
        self.W_rv = nn.Parameter(torch.empty(feat_size, hidden_dim).uniform_(-0.1, 0.1))
        self.W_id = nn.Parameter(torch.empty(hidden_dim, hidden_dim).uniform_(-0.1, 0.1))
        self.h = nn.Parameter(torch.empty(hidden_dim, 1).uniform_(-0.1, 0.1))
        self.b_1 = nn.Parameter(torch.empty(hidden_dim).fill_(0.1))
        self.b_2 = nn.Parameter(torch.empty(1).fill_(0.1))
        self.ebd_vals = nn.Embedding(vocab_size, hidden_dim, padding_idx=padding_idx)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, feat, other_id):
        """feat [bz, dnum, hidden], other_id [bz, dnum] -> (out [bz, hidden], att_scores [bz, dnum, 1])."""
        drop = RF.dropout_multiplier((feat.shape[0], feat.shape[2]), self.dropout.p, self.training, feat.device)
        return RF.review_attention(feat, other_id, self.W_rv, self.W_id, self.h, self.b_1, self.b_2,
                                   self.ebd_vals.weight, pad_idx=self.padding_idx, drop=drop)


class NARRE(nn.Module):
    def __init__(self, user_size, item_size, vocab_size, kernel_sizes, hidden_dim, embedding_dim, att_dim, latent_dim,
                 max_doc_num, max_doc_len, dropout, word_padding_idx, user_padding_idx, item_padding_idx,
                 pretrained_embeddings, arch):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.hiddem_dim = hidden_dim      # (sic) attribute name of the reference
        self.doc_num = max_doc_num
        self.doc_len = max_doc_len

        # narre.py:151 does not forward word_padding_idx to WordEmbedding: padding_idx stays 0
        self.word_embeddings = WordEmbedding(vocab_size, embedding_dim, pretrained_embeddings=pretrained_embeddings)
        self.ngram = NgramFeat(kernel_sizes, embedding_dim, hidden_dim, max_doc_len, arch=arch)
        self.user_att = LinearAttention(item_size, hidden_dim, att_dim, dropout, padding_idx=item_padding_idx)
        self.item_att = LinearAttention(user_size, hidden_dim, att_dim, dropout, padding_idx=user_padding_idx)
        self.user_feat = LastFeat(user_size, hidden_dim, latent_dim, padding_idx=user_padding_idx)
        self.item_feat = LastFeat(item_size, hidden_dim, latent_dim, padding_idx=item_padding_idx)
        self.fm = FM(user_size, item_size, latent_dim, dropout, user_padding_idx=user_padding_idx,
                     item_padding_idx=item_padding_idx)
        self.user_size, self.item_size, self.vocab_size = user_size, item_size, vocab_size
        self.validate_ids = True      # device-side range check of every id tensor (functional.sanitize_ids), see DeepCoNNpp

    def forward(self, u_text, i_text, u_text_masks, i_text_masks, u_id, i_id, reuid, reiid):
        """u_text/i_text [bz, doc_num, doc_len] int64, masks same shape bool, u_id/i_id [bz],
        reuid/reiid [bz, doc_num]  ->  (pred [bz], u_att [bz, doc_num, 1], i_att [bz, doc_num, 1]).

        Reviews are folded into the batch (narre.py:170-176); both sides go through ONE launch of the
        fused gather+conv+pool kernel as 2*bz*doc_num documents of doc_len tokens."""
        bz = u_text.shape[0]
        R, T = self.doc_num, self.doc_len
        if self.validate_ids:
            wp = self.word_embeddings.padding_idx
            ids, u_id, i_id, reuid, reiid = RF.sanitize_ids(
                [(u_text.reshape(-1, T), self.vocab_size, wp), (i_text.reshape(-1, T), self.vocab_size, wp),
                 (u_id, self.user_size, self.user_feat.padding_idx), (i_id, self.item_size, self.item_feat.padding_idx),
                 (reuid, self.item_size, self.user_att.padding_idx), (reiid, self.user_size, self.item_att.padding_idx)],
                stack_first_two=True)
        else:
            ids = RF.stack_rows(u_text.reshape(-1, T), i_text.reshape(-1, T))
        masks = RF.stack_rows(u_text_masks.reshape(-1, T), i_text_masks.reshape(-1, T))
        feats = self.ngram.encode(self.word_embeddings.weight, ids, masks, padding_idx=self.word_embeddings.padding_idx)
        # both attention pools (user_att keyed by item ids, item_att by user ids: narre.py:177-178) in ONE launch per stage: the conv
        # output is already the stacked [2, bz, R, H] block, the pooled features go to the rating head as one [2*bz, H] block, and
        # so do the gradients on the way back -- no unbind / stack, no second stream
        ua, ia = self.user_att, self.item_att
        other = RF.stack_rows(reuid, reiid).view(2, bz, R)          # a view when the two id tensors are neighbours (sanitize_ids, the step's input block)
        drop = RF.dropout_multiplier((2 * bz, self.hiddem_dim), ua.dropout.p, ua.training, feats.device)
        out, att = RF.review_attention2(
            feats.view(2, bz, R, self.hiddem_dim), other,
            (ua.W_rv, ua.W_id, ua.h, ua.b_1, ua.b_2, ua.ebd_vals.weight), (ia.W_rv, ia.W_id, ia.h, ia.b_1, ia.b_2, ia.ebd_vals.weight),
            pad_idx=(ua.padding_idx, ia.padding_idx), drop=None if drop is None else drop.view(2, bz, self.hiddem_dim))
        u_att_scores, i_att_scores = att[0], att[1]

        pred = rating_head(self.user_feat, self.item_feat, self.fm, out.view(2 * bz, self.hiddem_dim), None, u_id, i_id)
        return pred.view(-1), u_att_scores, i_att_scores
