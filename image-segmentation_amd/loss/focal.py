"""Multi-class focal loss on logits with the reference's call signature
(/root/reference/loss/focal.py: focal_loss_with_logits :108-137).  CUDA tensors run the fused HIP
kernels of segk.losses (shared softmax with the Dice sums); CPU tensors the same formula in torch ops."""
from segk.losses import focal_loss_with_logits  # noqa: F401
