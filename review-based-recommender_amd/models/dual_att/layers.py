"""D-ATT layers with the reference's names, constructor arguments and parameter names
(models/dual_att/layers.py:8-89), backed by the HIP kernels of csrc/.

nn.Conv1d / nn.Sequential objects are kept purely as parameter containers so that the state_dict
keys (`attn.0.weight`, `conv.0.bias`, `conv1.0.weight`, ...) match the reference; no ATen conv runs.
The modules take TOKEN IDS plus the shared word table (`encode`), because the gather is fused into
the kernels; `forward(x)` with an embedded tensor is served by addressing x as a table."""
import torch
import torch.nn as nn

from ... import functional as RF
from ..deepconn.layers import WordEmbedding as _WordEmbedding


class WordEmbedding(_WordEmbedding):
    """dual_att/layers.py:8-23 (adds the `sparse` flag; sparse gradients are not produced by the HIP path)."""

    def __init__(self, vocab_size, embedding_dim, pretrained_embeddings=None, padding_idx=0, freeze_embeddings=False,
                 sparse=False):
        super().__init__(vocab_size, embedding_dim, pretrained_embeddings, padding_idx, freeze_embeddings)
        if sparse:
            raise RuntimeError("sparse embedding gradients are not supported by the HIP path")


def _as_table(x):
    """Address an embedded tensor [bz, L, E] as a table with ids = arange (no padding row)."""
    bz, L, E = x.shape
    ids = torch.arange(bz * L, device=x.device, dtype=torch.int64).view(bz, L)
    return x.reshape(bz * L, E), ids


class LocalAttention(nn.Module):
    """layers.py:25-53: per-token sigmoid gate from a win-wide conv (E -> 1, 'same'), gated 1x1 conv (E -> out),
    tanh, max over all doc_len positions."""

    def __init__(self, doc_len, window_size, out_size, emb_size=100):
        super().__init__()
        self.window_size = window_size
        self.doc_len = doc_len
        self.out_size = out_size
        self.emb_size = emb_size
        self.padding_size = (self.window_size - 1) // 2
        self.attn = nn.Sequential(nn.Conv1d(emb_size, 1, kernel_size=window_size, padding=self.padding_size), nn.Sigmoid())
        self.conv = nn.Sequential(nn.Conv1d(emb_size, out_size, kernel_size=1), nn.Tanh(), nn.MaxPool1d(doc_len))

    def encode(self, table, ids, padding_idx=0, rows=None):
        """`table`: the word table, or a pair of its aliases (RF.table_fanout) for the gate and the conv."""
        t_gate, t_conv = table if isinstance(table, (tuple, list)) else (table, table)
        gate = RF.datt_gate(t_gate, self.attn[0].weight, self.attn[0].bias, ids, is_global=False, padding_idx=padding_idx,
                            rows=rows)
        c = self.conv[0]
        # pad_runs: the gate of a position is a function of the 5 tokens around it, so positions deep inside a run of padding
        # all see one gate value and one table row -- the conv encodes such runs once (exact)
        return RF.textcnn(t_conv, ids, None, [c.weight], [c.bias], gate=gate, pad_mode=RF.PAD_SAME, act=RF.ACT_TANH,
                          padding_idx=padding_idx, pad_runs=self.window_size <= 17)      # [bz, out_size]

    def forward(self, x):
        table, ids = _as_table(x)
        return self.encode(table, ids, padding_idx=None).unsqueeze(-1)  # [bz, out_size, 1]


class GlobalAttention(nn.Module):
    """layers.py:55-89: ONE sigmoid scalar per document (conv kernel = doc_len), gated 'valid' convs k = 2, 3, 4
    (hard-coded in the reference), tanh, max over doc_len - k + 1 positions."""

    def __init__(self, doc_len, out_size, emb_size=100):
        super().__init__()
        self.doc_len = doc_len
        self.out_size = out_size
        self.emb_size = emb_size
        self.attn = nn.Sequential(nn.Conv1d(emb_size, 1, kernel_size=doc_len), nn.Sigmoid())
        self.conv1 = nn.Sequential(nn.Conv1d(emb_size, out_size, kernel_size=2), nn.Tanh(), nn.MaxPool1d(doc_len - 1))
        self.conv2 = nn.Sequential(nn.Conv1d(emb_size, out_size, kernel_size=3), nn.Tanh(), nn.MaxPool1d(doc_len - 2))
        self.conv3 = nn.Sequential(nn.Conv1d(emb_size, out_size, kernel_size=4), nn.Tanh(), nn.MaxPool1d(doc_len - 3))

    def encode(self, table, ids, padding_idx=0, rows=None):
        """`table`: the word table, or a pair of its aliases (RF.table_fanout) for the gate and the convs."""
        t_gate, t_conv = table if isinstance(table, (tuple, list)) else (table, table)
        gate = RF.datt_gate(t_gate, self.attn[0].weight, self.attn[0].bias, ids, is_global=True, padding_idx=padding_idx,
                            rows=rows)
        convs = [self.conv1[0], self.conv2[0], self.conv3[0]]
        # the three widths run in ONE launch of the fused kernel; channels come back width-major
        # pad_runs: the global gate is one scalar per document (exactly uniform over its padding)
        return RF.textcnn(t_conv, ids, None, [c.weight for c in convs], [c.bias for c in convs], gate=gate,
                          pad_mode=RF.PAD_VALID, act=RF.ACT_TANH, padding_idx=padding_idx, pad_runs=True)   # [bz, 3*out_size]

    def forward(self, x):
        table, ids = _as_table(x)
        out = self.encode(table, ids, padding_idx=None)
        o = self.out_size
        return out[:, :o].unsqueeze(-1), out[:, o:2 * o].unsqueeze(-1), out[:, 2 * o:].unsqueeze(-1)
