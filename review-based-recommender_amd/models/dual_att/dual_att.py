"""D-ATT (dual local/global attention) with the reference's constructor / forward signature and
state_dict keys (models/dual_att/dual_att.py:19-61), running on the HIP kernels of csrc/.

The reference file's legacy `__main__` block (dual_att.py:63-150; undefined ReviewData / CNNDLGA) and its
unused nltk / pandas imports are not part of the model and have no counterpart here."""
import os

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
        convs = [local.conv[0], glob.conv1[0], glob.conv2[0], glob.conv3[0]]
        weights = [c.weight for c in convs]
        if (local.conv[0].kernel_size[0] == 1 and os.environ.get("RBR_DATT_MERGED", "1") != "0"
                and RF.textcnn_product_applies(tabs[1], docs, weights, RF.PAD_VALID, RF.ACT_TANH, pad)):
            # The tower's two gated convs read the same tokens: as ONE conv call of four banks -- the 1-wide local conv ('same'
            # = 'valid' at width 1) under the local gate, the 2/3/4-wide global convs under the global gate
            # (RBR_CONV_GATE_SPLIT) -- they share the token list, the product-table GEMM, the gather launch, and in the
            # backward G and the sparse product.  Same results (layers.py:43-53,81-89); channels come back local first.
            gate_l = RF.datt_gate(tabs[0], local.attn[0].weight, local.attn[0].bias, docs, is_global=False, padding_idx=pad,
                                  rows=rows)
            gate_g = RF.datt_gate(tabs[2], glob.attn[0].weight, glob.attn[0].bias, docs, is_global=True, padding_idx=pad,
                                  rows=rows)
            feat = RF.textcnn(tabs[1], docs, None, weights, [c.bias for c in convs], gate=(gate_l, gate_g), gate_split=1,
                              pad_mode=RF.PAD_VALID, act=RF.ACT_TANH, padding_idx=pad, pad_runs=local.window_size <= 17)
            return feat, None                                   # [bz, l_out + 3 * g_out]: cat((local, global), 1) already
        return local.encode(tabs[0:2], docs, pad, rows=rows), glob.encode(tabs[2:4], docs, pad, rows=rows)

    def _fc(self, feat):
        p = self.fc[2].p
        drop = RF.dropout_multiplier((feat.shape[0], self.fc[0].out_features), p, self.training, feat.device)
        hid = RF.linear(feat, self.fc[0].weight, self.fc[0].bias, relu=True, drop=drop)
        return RF.linear(hid, self.fc[3].weight, self.fc[3].bias)

    def _tower_params(self, local, glob):
        convs = [local.conv[0], glob.conv1[0], glob.conv2[0], glob.conv3[0]]
        return (local.attn[0].weight, local.attn[0].bias, glob.attn[0].weight, glob.attn[0].bias,
                [c.weight for c in convs], [c.bias for c in convs])

    def forward(self, u_docs, i_docs):
        """u_docs / i_docs [bz, doc_len] int64 -> ratings [bz]."""
        bz = u_docs.shape[0]
        pad = self.word_embeddings.padding_idx
        if self.validate_ids:
            u_docs, i_docs = RF.sanitize_ids([(u_docs, self.vocab_size, pad), (i_docs, self.vocab_size, pad)])
        # Both towers through every kernel in ONE pass (RF.datt_towers): the towers run the same ops on separate parameters and
        # documents of equal shapes (dual_att.py:45-57), so each kernel of the chain -- token rows, both gates, the merged
        # four-bank conv, and their backwards -- is launched once with gridDim.z = 2 instead of once per tower.
        u_par, i_par = self._tower_params(self.u_local_atten, self.u_global_atten), self._tower_params(self.i_local_atten, self.i_global_atten)
        if u_docs.shape == i_docs.shape and self.u_local_atten.window_size == self.i_local_atten.window_size:
            docs2 = RF.stack_rows(u_docs, i_docs)
            if RF.datt_pair_applies(self.word_embeddings.weight, docs2, u_par[0], u_par[4]):
                feats = RF.datt_towers(self.word_embeddings.weight, docs2, u_par, i_par, padding_idx=pad,
                                       pad_runs=self.u_local_atten.window_size <= 17)
                return RF.pair_dot(self._fc(feats)).view(-1)                # sum(u_feat * i_feat, 1)  (dual_att.py:58)
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
        if u_glo is None and i_glo is None:
            feats = self._fc(RF.stack_rows(u_loc, i_loc))                   # the merged conv's rows are the towers' cat already
        else:
            if u_glo is None:
                u_loc, u_glo = u_loc[:, :self.u_local_atten.out_size], u_loc[:, self.u_local_atten.out_size:]
            if i_glo is None:
                i_loc, i_glo = i_loc[:, :self.i_local_atten.out_size], i_loc[:, self.i_local_atten.out_size:]
            feats = self._fc(RF.block_cat(u_loc, u_glo, i_loc, i_glo))      # [2*bz, hidden_2], user rows first
        return RF.pair_dot(feats).view(-1)                                  # sum(u_feat * i_feat, 1)  (dual_att.py:58)
