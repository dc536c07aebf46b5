"""Host restatement of the masked MSE used by the Noise2Void pipeline (reference: utils/n2v_loss.py:6-17).  The device
path is SSDN_OP_MASK_MSE; this function exists for API parity (ssdn.utils.n2v_loss.loss_mask_mse)."""
from torch import Tensor


def loss_mask_mse(masked_coords: Tensor, input: Tensor, target: Tensor) -> Tensor:
    """[B,C] sum over the mask coordinates OF BATCH ELEMENT 0 (reference quirk)."""
    c = masked_coords[0].long()
    diff = target[:, :, c[:, 0], c[:, 1]] - input[:, :, c[:, 0], c[:, 1]]
    return (diff ** 2).sum(-1)
