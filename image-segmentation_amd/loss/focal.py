"""Multi-class focal loss on logits with the reference's call signature
(/root/reference/loss/focal.py: reduce :5-30, _general_focal_loss :32-75, _focal_loss :77-106, focal_loss_with_logits :108-137).  CUDA tensors run the fused HIP
kernels of segk.losses (shared softmax with the Dice sums); CPU tensors the same formula in torch ops."""
from segk.losses import _focal_loss, _general_focal_loss, focal_loss_with_logits, reduce  # noqa: F401
