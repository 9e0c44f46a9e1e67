"""Mirror of models/deepconn/utils.py:49-61 for HIP tensors."""
import torch


def masked_tensor(inputs: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """Zero the rows of `inputs` [*, hidden] whose mask [*] is False (utils.py:49-61).

    Inside the models this never runs: the mask is applied while the token rows are gathered
    into LDS (csrc/textcnn_fwd.hip).  Kept for callers that use the helper directly."""
    assert inputs.shape[:-1] == masks.shape
    return inputs * masks.unsqueeze(-1).to(inputs.dtype)
