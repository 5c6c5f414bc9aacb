"""segk: MI355X-native kernels and runtime of the segmentation training step."""
from . import _lib  # noqa: F401
from ._lib import SegkError, available  # noqa: F401
