"""Edge-aware Dice loss with the reference's call signatures (/root/reference/loss/hausdorff.py: get_edg_c :71-108,
edge_conv_loss :110-128).  CUDA tensors run the HIP edge kernels and the fused Dice kernels of segk.losses; CPU tensors
the same formulas in torch ops.  The Hausdorff distance helpers of that file (numpy / scikit-image code on the host) are
not part of the training step and are not provided."""
from segk.losses import edge_conv_loss, get_edg_c  # noqa: F401
