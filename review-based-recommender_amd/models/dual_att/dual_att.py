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

        self.word_embeddings = WordEmbedding(vocab_size, emb_size, pretrained_embeddings=pretrained_embeddings)
        self.u_local_atten = LocalAttention(doc_len, l_window_size, l_out_size, emb_size)
        self.u_global_atten = GlobalAttention(doc_len, g_out_size, emb_size)
        self.i_local_atten = LocalAttention(doc_len, l_window_size, l_out_size, emb_size)
        self.i_global_atten = GlobalAttention(doc_len, g_out_size, emb_size)
        # ONE fc shared by both towers (dual_att.py:31,51,57); kept as nn.Sequential for the state_dict keys fc.0 / fc.3
        self.fc = nn.Sequential(nn.Linear(self.fc_input, hidden_size_1), nn.ReLU(), nn.Dropout(dropout),
                                nn.Linear(hidden_size_1, hidden_size_2))

    def _tower(self, docs, local, glob):
        table = self.word_embeddings.weight
        pad = self.word_embeddings.padding_idx
        feat = torch.cat((local.encode(table, docs, pad), glob.encode(table, docs, pad)), dim=1)   # [bz, fc_input]
        p = self.fc[2].p
        drop = RF.dropout_multiplier((feat.shape[0], self.fc[0].out_features), p, self.training, feat.device)
        hid = RF.linear(feat, self.fc[0].weight, self.fc[0].bias, relu=True, drop=drop)
        return RF.linear(hid, self.fc[3].weight, self.fc[3].bias)

    def forward(self, u_docs, i_docs):
        """u_docs / i_docs [bz, doc_len] int64 -> ratings [bz]."""
        u_feat = self._tower(u_docs, self.u_local_atten, self.u_global_atten)
        i_feat = self._tower(i_docs, self.i_local_atten, self.i_global_atten)
        ratings = torch.sum(torch.mul(u_feat, i_feat), 1)
        return ratings.view(-1)
