"""Soft (squared) Dice loss and arg-max Dice metric with the reference's call signatures
(/root/reference/loss/dice.py: dice_loss_sq :4-23, dice_coefficient_sq :25-87, dice_coefficient :89-184,
dice_coefficientold :186-256, dice_coefficient_sa :275-325, dice_loss_sa :327-345, logcoshdiceloss :347-349).
CUDA tensors run the fused HIP kernels of segk.losses; CPU tensors the same formulas in torch ops."""
from segk.losses import (dice_coefficient, dice_coefficient_sa, dice_coefficient_sq, dice_coefficientold, dice_loss_sa,  # noqa: F401
                         dice_loss_sq, logcoshdiceloss)
