"""Synthetic stand-ins for the reference's dataset readers (the real ones read private MRI volumes)."""
