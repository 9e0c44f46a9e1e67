"""D-ATT (dual local/global attention) with the reference's constructor / forward signature and
state_dict keys (models/dual_att/dual_att.py:19-61), running on the HIP kernels of csrc/.

The reference file's legacy `__main__` block (dual_att.py:63-150; undefined ReviewData / CNNDLGA) and its
unused nltk / pandas imports are not part of the model and have no counterpart here."""
import torch
import torch.nn as nn

from ... import functional as RF
from .layers import GlobalAttention, LocalAttention, WordEmbedding


class DualAtt(nn.Module):
    def __init__(self, vocab_size, doc_len, l_window_size=5, l_out_size=200, g_out_size=100, emb_size=100,
                 hidden_size_1=500, hidden_size_2=50, dropout=0.5, pretrained_embeddings=None):
        super().__init__()
        self.fc_input = l_out_size + 3 * g_out_size
        self.vocab_size = vocab_size
        self.validate_ids = True      # device-side range check of the token ids (functional.sanitize_ids), see DeepCoNNpp

        self.word_embeddings = WordEmbedding(vocab_size, emb_size, pretrained_embeddings=pretrained_embeddings)
        self.u_local_atten = LocalAttention(doc_len, l_window_size, l_out_size, emb_size)
        self.u_global_atten = GlobalAttention(doc_len, g_out_size, emb_size)
        self.i_local_atten = LocalAttention(doc_len, l_window_size, l_out_size, emb_size)
        self.i_global_atten = GlobalAttention(doc_len, g_out_size, emb_size)
        # ONE fc shared by both towers (dual_att.py:31,51,57); kept as nn.Sequential for the state_dict keys fc.0 / fc.3
        self.fc = nn.Sequential(nn.Linear(self.fc_input, hidden_size_1), nn.ReLU(), nn.Dropout(dropout),
                                nn.Linear(hidden_size_1, hidden_size_2))

    def _encode(self, docs, local, glob, tabs):
        """`tabs`: four aliases of the word table (RF.table_fanout), one per table-consuming op of the tower."""
        pad = self.word_embeddings.padding_idx
        rows = RF.datt_token_rows(docs, tabs[0].shape[0])      # distinct-token maps, once per tower: both gates work on them
        return local.encode(tabs[0:2], docs, pad, rows=rows), glob.encode(tabs[2:4], docs, pad, rows=rows)

    def _fc(self, feat):
        p = self.fc[2].p
        drop = RF.dropout_multiplier((feat.shape[0], self.fc[0].out_features), p, self.training, feat.device)
        hid = RF.linear(feat, self.fc[0].weight, self.fc[0].bias, relu=True, drop=drop)
        return RF.linear(hid, self.fc[3].weight, self.fc[3].bias)

    def forward(self, u_docs, i_docs):
        """u_docs / i_docs [bz, doc_len] int64 -> ratings [bz]."""
        bz = u_docs.shape[0]
        if self.validate_ids:
            pad = self.word_embeddings.padding_idx
            u_docs, i_docs = RF.sanitize_ids([(u_docs, self.vocab_size, pad), (i_docs, self.vocab_size, pad)])
        # eight ops of the step produce gradient for the word table (a gate and a conv, local and global, per tower): each gets
        # its own alias, and their backwards add their rows into one buffer instead of eight dense gradients summed by autograd
        # (Tried in round 3 and dropped: the item tower on a stream of its own, so that one tower's ~40 latency-bound launches run
        # under the other's long kernels -- autograd then replays every backward op on its forward's stream, and recording that
        # backward into a hipGraph crashed the process inside the capture; the towers stay one after the other.)
        tabs = RF.table_fanout(self.word_embeddings.weight, 8)
        u_loc, u_glo = self._encode(u_docs, self.u_local_atten, self.u_global_atten, tabs[0:4])
        i_loc, i_glo = self._encode(i_docs, self.i_local_atten, self.i_global_atten, tabs[4:8])
        # the fc is ONE module shared by both towers (dual_att.py:31,51,57): both sides go through its two GEMMs (and their
        # backward) as a single 2*bz batch, user rows first
        # cat((local, global), 1) per tower (dual_att.py:50,56), user rows over item rows: one launch (RF.block_cat)
        feats = self._fc(RF.block_cat(u_loc, u_glo, i_loc, i_glo))          # [2*bz, hidden_2], user rows first
        return RF.pair_dot(feats).view(-1)                                  # sum(u_feat * i_feat, 1)  (dual_att.py:58)
