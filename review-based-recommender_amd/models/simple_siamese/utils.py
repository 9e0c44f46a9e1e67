"""Helpers of models/simple_siamese/utils.py that the model file imports (mask plumbing only)."""
import torch


def get_rev_mask(inputs: torch.Tensor) -> torch.Tensor:
    """utils.py:94-106 -- a review whose tokens are all 0 (pad) is masked out.  inputs [bz, rv_num, rv_len] -> bool [bz, rv_num]."""
    return inputs.sum(dim=-1) != 0
