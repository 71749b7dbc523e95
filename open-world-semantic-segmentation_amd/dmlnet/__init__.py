"""Runtime under the reference-shaped `network` / `utils` packages: C-ABI binding, static-plan executor,
fused optimizer and the RCCL gradient reducer."""
from ._lib import DmlError, LIB_PATH, load  # noqa: F401
