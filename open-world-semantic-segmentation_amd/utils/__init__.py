"""Drop-in for the parts of DeepLabV3Plus-Pytorch/utils/ the embedding drivers use
(utils/__init__.py there re-exports loss, scheduler and the BN helpers)."""
from .loss import CrossEntropyLoss, DMLLoss  # noqa: F401
from .scheduler import PolyLR  # noqa: F401
from .misc import set_bn_momentum, fix_bn, mkdir  # noqa: F401
from .scores import argmax_msp, dissum_score, novel_relabel, mean_prototype, extract_prototype  # noqa: F401
from . import ext_transforms  # noqa: F401,E402
