"""Drop-in for DeepLabV3Plus-Pytorch/network/__init__.py:1-2 of the reference."""
from .modeling import *  # noqa: F401,F403
from .modeling import convert_to_separable_conv  # noqa: F401
