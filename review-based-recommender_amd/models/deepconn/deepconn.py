"""DeepCoNN++ with the reference's constructor / forward signature and state_dict keys
(models/deepconn/deepconn.py:10-53), running on the HIP kernels of csrc/."""
import torch
import torch.nn as nn

from ... import functional as RF
from .layers import FM, LastFeat, NgramFeat, WordEmbedding, rating_head


class DeepCoNNpp(nn.Module):
    def __init__(self, user_size, item_size, vocab_size, kernel_sizes, embedding_dim, hidden_dim, latent_dim, doc_len,
                 pretrained_embeddings, dropout, arch="CNN"):
        super().__init__()
        self.user_size = user_size
        self.item_size = item_size
        self.vocab_size = vocab_size
        self.hidden_dim = hidden_dim

        self.word_embeddings = WordEmbedding(vocab_size, embedding_dim, pretrained_embeddings=pretrained_embeddings)
        self.ngram = NgramFeat(kernel_sizes, embedding_dim, hidden_dim, doc_len, arch=arch)
        self.user_feat = LastFeat(user_size, hidden_dim, latent_dim, padding_idx=0)
        self.item_feat = LastFeat(item_size, hidden_dim, latent_dim, padding_idx=0)
        self.fm = FM(user_size, item_size, latent_dim, dropout, user_padding_idx=0, item_padding_idx=0)
        # encode each distinct user / item document of the batch once.  Valid when a
        # document is a function of its id (the doc split builds one document per user / item); the reference
        # re-encodes the same document for every pair it appears in (deepconn.py:46-47).
        self.dedup_by_id = False
        # every id tensor passes a device-side range check first (one launch; functional.sanitize_ids): an id outside its
        # table becomes the padding row and functional.check_id_errors() raises nn.Embedding's IndexError at the next
        # synchronisation point.  False: the caller guarantees the ranges (a dataset validated at load time).
        self.validate_ids = True

    def _fused_ok(self, u_revs, i_revs) -> bool:
        if self.ngram.arch != "CNN" or u_revs.shape != i_revs.shape or u_revs.dim() != 2:
            return False
        conv = self.ngram.feature_layer[0]
        return RF.encode_head_applicable(self.word_embeddings.weight, 2 * u_revs.shape[0], u_revs.shape[1], conv.kernel_sizes,
                                         [conv.out_features_per_kz] * len(conv.kernel_sizes), self.word_embeddings.padding_idx)

    def forward(self, u_revs, i_revs, u_rev_masks, i_rev_masks, u_ids, i_ids):
        """u_revs/i_revs [bz, doc_len] int64, masks [bz, doc_len] bool, ids [bz] -> preds [bz].

        Both towers share the table and the TextCNN (deepconn.py:43-47), so the user and item
        documents go through ONE launch of the fused gather+conv+pool kernel as a 2*bz batch."""
        bz = u_revs.shape[0]
        stacked = None
        if self._fused_ok(u_revs, i_revs):
            # encoder + rating head as ONE autograd function (functional.encode_head): the id check rides in the conv's
            # prepare launch, the pool epilogue and -- inside train_step's fused_loss -- the MSE ride in the head launch
            pad = self.word_embeddings.padding_idx
            conv = self.ngram.feature_layer[0]
            uf, itf, fm = self.user_feat, self.item_feat, self.fm
            training = fm.training and torch.is_grad_enabled()
            if training and fm.dropout.p < 1.0:
                drop = float(fm.dropout.p)
            else:
                drop = RF.dropout_multiplier((bz, fm.h.shape[0]), fm.dropout.p, fm.training, u_revs.device)
            head = (uf.W, uf.b, uf.ebd.weight, itf.W, itf.b, itf.ebd.weight, fm.h, fm.g_bias, fm.user_bias.weight, fm.item_bias.weight)
            masks = RF.stack_rows(u_rev_masks, i_rev_masks)
            first = None
            if self.dedup_by_id:
                # first occurrence of every user / item id of the batch (two short launches, static shapes, no sync; ids outside
                # their table count as unique): the repeated documents' masks are blanked -- the encoder skips them -- and the
                # head launch reads the first occurrence's pooled features in their place
                first, masks = RF.dedup_rows(u_ids, i_ids, self.user_size, self.item_size, masks, u_revs.shape[1])
            if self.validate_ids:
                sets = [(u_revs, self.vocab_size, pad), (i_revs, self.vocab_size, pad), (u_ids, self.user_size, 0),
                        (i_ids, self.item_size, 0)]
                preds = RF.encode_head(self.word_embeddings.weight, None, masks, None, None, conv.weights(), conv.biases(), head,
                                       id_sets=sets, drop=drop, padding_idx=pad, pad_u=fm.user_padding_idx, pad_i=fm.item_padding_idx,
                                       first=first)
            else:
                preds = RF.encode_head(self.word_embeddings.weight, RF.stack_rows(u_revs, i_revs), masks, u_ids, i_ids,
                                       conv.weights(), conv.biases(), head, drop=drop, padding_idx=pad,
                                       pad_u=fm.user_padding_idx, pad_i=fm.item_padding_idx, first=first)
            return preds.view(bz)
        if self.validate_ids:
            pad = self.word_embeddings.padding_idx
            stacked, u_ids, i_ids = RF.sanitize_ids([(u_revs, self.vocab_size, pad), (i_revs, self.vocab_size, pad),
                                                     (u_ids, self.user_size, 0), (i_ids, self.item_size, 0)], stack_first_two=True)
            u_revs, i_revs = stacked[:bz], stacked[bz:]
        if self.dedup_by_id:
            # first occurrence of every id on the device (static shapes, no sync: the step stays graph-capturable); the
            # repeated documents are blanked, so the encoder skips them, and their features are row gathers
            ids = stacked if stacked is not None else RF.stack_rows(u_revs, i_revs)
            first, masks = RF.dedup_rows(u_ids, i_ids, self.user_size, self.item_size,
                                         RF.stack_rows(u_rev_masks, i_rev_masks), ids.shape[1])
            feats = self.ngram.encode(self.word_embeddings.weight, ids, masks, padding_idx=self.word_embeddings.padding_idx)
            u_rev_feats, i_rev_feats = feats.index_select(0, first), None       # [2*bz, H], user rows first
        else:
            ids = stacked if stacked is not None else RF.stack_rows(u_revs, i_revs)
            masks = RF.stack_rows(u_rev_masks, i_rev_masks)
            feats = self.ngram.encode(self.word_embeddings.weight, ids, masks, padding_idx=self.word_embeddings.padding_idx)
            u_rev_feats, i_rev_feats = feats, None          # [2*bz, H], user rows first: the head takes it whole
        preds = rating_head(self.user_feat, self.item_feat, self.fm, u_rev_feats, i_rev_feats, u_ids, i_ids)
        return preds.view(bz)
