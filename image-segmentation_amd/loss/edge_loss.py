"""One-hot encoding helper with the reference's call signature (/root/reference/loss/edge_loss.py: get_hot_enc :32-55)."""
from segk.losses import get_hot_enc  # noqa: F401
