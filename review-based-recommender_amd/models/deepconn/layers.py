"""Layer classes with the reference's names, constructor arguments, parameter names and
shapes (models/deepconn/layers.py), backed by the HIP kernels of csrc/.

The classes own ordinary nn.Parameters so optimisers, clip_grad_norm_, state_dict and
.to(device) behave exactly as with the reference modules.  Compute happens in
review_based_recommender_amd.functional (no ATen conv / embedding / matmul on the path).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import functional as RF


def _new_param(*shape) -> nn.Parameter:
    """Uninitialised fp32 parameter; the owning module's reset_parameters() fills it."""
    return nn.Parameter(torch.empty(*shape))


def _uniform_(t: torch.Tensor, bound: float) -> None:
    with torch.no_grad():
        t.uniform_(-bound, bound)


class WordEmbedding(nn.Module):
    """layers.py:9-24: nn.Embedding(V, D, padding_idx) table; optional pretrained rows; freeze flag."""

    def __init__(self, vocab_size, embedding_dim, pretrained_embeddings=None, padding_idx=0, freeze_embeddings=False):
        super().__init__()
        self.freeze_embeddings, self.padding_idx = freeze_embeddings, padding_idx
        # nn.Embedding is kept as the parameter container (state_dict key `embedding.weight`,
        # default N(0,1) init with a zero pad row); its forward is never used.
        self.embedding = nn.Embedding(vocab_size, embedding_dim, padding_idx=padding_idx)
        self.embedding.weight.requires_grad_(not freeze_embeddings)
        if pretrained_embeddings is None:
            print("[Warning] not use pretrained embeddings ...")          # the reference's message, layers.py:20
        else:
            self.embedding.load_state_dict({"weight": torch.as_tensor(pretrained_embeddings)})

    @property
    def weight(self) -> torch.Tensor:
        return self.embedding.weight

    def forward(self, inputs):
        """Materialised lookup [*, D] for callers that want the rows themselves (HIP row gather); ids are range-checked on
        the device first (functional.sanitize_ids / check_id_errors stand in for nn.Embedding's IndexError)."""
        (inputs,) = RF.sanitize_ids([(inputs, self.embedding.num_embeddings, self.padding_idx)])
        return RF.embedding(self.embedding.weight, inputs, self.padding_idx)


class MyConv1d(nn.Module):
    """layers.py:26-60: one nn.Conv1d parameter set per (odd) kernel width, `"3,5,7"` strings allowed."""

    def __init__(self, kernel_sizes, in_features, out_features):
        super().__init__()
        widths = [int(k) for k in (kernel_sizes.split(",") if isinstance(kernel_sizes, str) else kernel_sizes)]
        per_width, rest = divmod(out_features, len(widths))
        # the reference asserts both (layers.py:38-39): same exception type for callers that catch it
        if rest != 0:
            raise AssertionError("out_features must divide evenly over the kernel sizes")
        if any(k % 2 == 0 for k in widths):
            raise AssertionError("kernel sizes must be odd ('same' padding)")
        self.kernel_sizes, self.out_features_per_kz = widths, per_width
        # nn.Conv1d modules are the parameter containers (keys list_of_conv1d.{i}.weight / .bias, torch's default init)
        self.list_of_conv1d = nn.ModuleList(nn.Conv1d(in_features, per_width, k, padding=k // 2) for k in widths)

    def weights(self):
        return [c.weight for c in self.list_of_conv1d]

    def biases(self):
        return [c.bias for c in self.list_of_conv1d]

    def forward(self, inputs):
        """Reference signature (layers.py:46-60): inputs [bz, in_features, seq_len] (N x C x L) -> [bz, out_features, seq_len],
        every width zero-padded 'same', channels width-major.  The models never call this (their conv is fused with the
        gather and the max-pool and never materialises a [bz, C, L] activation); it serves callers of the layer itself.
        The contraction runs on the HIP GEMM (functional.linear, forward and backward): T[n, (w, j, c)] = <x[n, :], W_w[c, :, j]>
        for every position n, then out[b, c, l] = bias[c] + sum_j T[(b, l + j - pad_w), (w, j, c)] (rbr_conv_shift_add_*).  What is
        left to torch is re-layout only: the N x C x L input as rows, and the weights as one [sum kz*ch, D] matrix."""
        bz, cin, L = inputs.shape
        x = inputs.transpose(1, 2).reshape(bz * L, cin)                       # one row per position
        wp = torch.cat([c.weight.permute(2, 0, 1).reshape(-1, cin) for c in self.list_of_conv1d], dim=0)   # [(w, j, c), D]
        t = RF.linear(x, wp, None)                                            # [bz * L, sum kz * ch] on the HIP GEMM
        # padding, the kz shifted adds, bias and the channel cat in the N x C x L layout: one HIP launch (forward) / two (backward)
        return RF.conv_shift_add(t, bz, L, self.kernel_sizes, [c.weight.shape[0] for c in self.list_of_conv1d], self.biases())


class HierPooling(nn.Module):
    """layers.py:62-98 parameter container: optional Linear(in->out) projection."""

    def __init__(self, in_features, out_features, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.proj_layer = nn.Linear(in_features, out_features) if in_features != out_features else None


class NgramFeat(nn.Module):
    """layers.py:100-136.  arch="CNN": MyConv1d -> ReLU -> MaxPool1d(seq_len) (all positions, pads included);
    arch="HierPooling": avg-pool window -> global max -> optional Linear -> ReLU."""

    def __init__(self, kernel_sizes, in_features, out_features, seq_len, dropout=0., arch="CNN"):
        super().__init__()
        self.arch = arch
        self.seq_len = seq_len
        if arch == "CNN":
            print("use CNN archiecture for Ngram.")
            self.feature_layer = nn.Sequential(MyConv1d(kernel_sizes, in_features, out_features), nn.ReLU(),
                                               nn.MaxPool1d(seq_len))
        elif arch == "HierPooling":
            print("use HierPooling arch for Ngram.")
            assert len(kernel_sizes) == 1
            self.feature_layer = nn.Sequential(HierPooling(in_features, out_features, kernel_sizes[0]), nn.ReLU())
        else:
            raise ValueError(f"{arch} is not predefined.")
        # stored, never applied -- as in the reference (layers.py:118-121, quirk 7)
        self.dropout = nn.Dropout(p=dropout) if dropout else None

    # fused entry used by the models: token ids in, pooled features out
    def encode(self, table: torch.Tensor, ids: torch.Tensor, masks: torch.Tensor, padding_idx=0) -> torch.Tensor:
        """ids/masks [n_docs, L] -> [n_docs, out_features]."""
        if self.arch == "CNN":
            conv = self.feature_layer[0]
            return RF.textcnn(table, ids, masks, conv.weights(), conv.biases(), padding_idx=padding_idx)
        hp = self.feature_layer[0]
        pw = hp.proj_layer.weight if hp.proj_layer is not None else None
        pb = hp.proj_layer.bias if hp.proj_layer is not None else None
        return RF.hier_pool(table, ids, masks, hp.kernel_size, pw, pb, padding_idx=padding_idx)

    def forward(self, inputs, input_masks):
        """Reference signature: inputs [bz, seq_len, in_features] (already embedded), masks [bz, seq_len]
        -> [bz, out_features, 1] for CNN / [bz, out_features] for HierPooling.

        Runs the same fused kernels by treating `inputs` as a [bz*seq_len, D] table addressed by
        ids = arange (no padding row), so gradients flow back into `inputs`."""
        bz, L, D = inputs.shape
        table = inputs.reshape(bz * L, D)
        ids = torch.arange(bz * L, device=inputs.device, dtype=torch.int64).view(bz, L)
        out = self.encode(table, ids, input_masks, padding_idx=None)
        return out.unsqueeze(-1) if self.arch == "CNN" else out


class LastFeat(nn.Module):
    """layers.py:138-165 parameters: W [feat, latent], b [latent], ebd [vocab, latent] (pad row re-initialised)."""

    BOUND = 0.1        # U(-0.1, 0.1) for W and the id embedding (pad row included, quirk 4)
    BIAS_INIT = 0.1    # deepconn / narre: b = 0.1; simple_siamese: 0 (its subclass)

    def __init__(self, vocab_size, feat_size, latent_dim, padding_idx):
        super().__init__()
        self.padding_idx = padding_idx
        self.W, self.b = _new_param(feat_size, latent_dim), _new_param(latent_dim)      # registered in this order
        self.ebd = nn.Embedding(vocab_size, latent_dim, padding_idx=padding_idx)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_(self.W, self.BOUND)
        _uniform_(self.ebd.weight, self.BOUND)
        with torch.no_grad():
            self.b.fill_(self.BIAS_INIT)

    def forward(self, text_feat, my_id):
        """Reference signature (layers.py:156-165): text_feat [bz, feat_size], my_id [bz] -> [bz, latent_dim] =
        text_feat @ W + b + ebd(my_id).  The models use the fused rating head (rating_head below); this serves callers of
        the layer itself: HIP GEMM + HIP row gather, one elementwise add."""
        (my_id,) = RF.sanitize_ids([(my_id, self.ebd.num_embeddings, self.padding_idx)])
        return RF.linear(text_feat, self.W.t(), self.b) + RF.embedding(self.ebd.weight, my_id, self.padding_idx)


class FM(nn.Module):
    """layers.py:167-209 parameters: h [latent,1], g_bias [1], user_bias [U,1], item_bias [I,1]; Dropout(p)."""

    BOUND = 0.1
    G_BIAS_INIT = 0.1      # deepconn / narre; simple_siamese starts the global bias at 4.0 (its subclass)
    WITH_ID_BIASES = True

    def __init__(self, user_size, item_size, latent_dim, dropout, user_padding_idx, item_padding_idx):
        super().__init__()
        self.user_padding_idx, self.item_padding_idx = user_padding_idx, item_padding_idx
        self.dropout = nn.Dropout(dropout)
        self.h = _new_param(latent_dim, 1)
        if self.WITH_ID_BIASES:        # registration order h, user_bias, item_bias, g_bias = the reference's state_dict order
            self.user_bias = nn.Embedding(user_size, 1, padding_idx=user_padding_idx)
            self.item_bias = nn.Embedding(item_size, 1, padding_idx=item_padding_idx)
        self.g_bias = _new_param(1)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_(self.h, self.BOUND)
        if self.WITH_ID_BIASES:
            _uniform_(self.user_bias.weight, self.BOUND)
            _uniform_(self.item_bias.weight, self.BOUND)
        with torch.no_grad():
            self.g_bias.fill_(self.G_BIAS_INIT)

    def forward(self, u_feat, i_feat, u_id, i_id):
        """Reference signature (layers.py:189-209): u_feat / i_feat [bz, latent_dim], ids [bz] -> pred [bz, 1] =
        dropout(relu(u_feat * i_feat)) @ h + user_bias(u_id) + item_bias(i_id) + g_bias.  The models use the fused rating
        head; this serves callers of the layer itself (HIP GEMM, HIP row gathers, HIP dropout multiplier)."""
        fm = torch.relu(u_feat * i_feat)
        drop = RF.dropout_multiplier(fm.shape, self.dropout.p, self.training, fm.device)
        if drop is not None:
            fm = fm * drop
        pred = RF.linear(fm, self.h.t(), None) + self.g_bias
        if self.WITH_ID_BIASES:
            u_id, i_id = RF.sanitize_ids([(u_id, self.user_bias.num_embeddings, self.user_padding_idx),
                                          (i_id, self.item_bias.num_embeddings, self.item_padding_idx)])
            pred = pred + RF.embedding(self.user_bias.weight, u_id, self.user_padding_idx) \
                        + RF.embedding(self.item_bias.weight, i_id, self.item_padding_idx)
        return pred


def rating_head(user_feat: LastFeat, item_feat: LastFeat, fm: FM, u_text_feat, i_text_feat, u_ids, i_ids):
    """LastFeat(user) + LastFeat(item) + FM in one HIP kernel pair (layers.py:156-165,189-209).
    i_text_feat None: u_text_feat holds both towers, [2*bz, H] with the user rows first."""
    if fm.training and fm.dropout.p < 1.0 and torch.is_grad_enabled():
        drop = float(fm.dropout.p)       # training forward: the head kernel draws the mask (and clears its gradient buffer) itself
    else:
        drop = RF.dropout_multiplier((u_ids.shape[0], fm.h.shape[0]), fm.dropout.p, fm.training, u_text_feat.device)
    return RF.pair_head(u_text_feat, i_text_feat, u_ids, i_ids,
                        user_feat.W, user_feat.b, user_feat.ebd.weight,
                        item_feat.W, item_feat.b, item_feat.ebd.weight,
                        fm.h, fm.g_bias, fm.user_bias.weight, fm.item_bias.weight,
                        drop=drop, pad_u=fm.user_padding_idx, pad_i=fm.item_padding_idx)
