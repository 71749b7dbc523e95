"""The anomaly sub-project's model API (reference: anomaly/models/__init__.py, models/models.py) on the MI355X path."""
from .models import ModelBuilder, SegmentationModuleOOD, SynchronizedBatchNorm2d, evaluate_multiscale  # noqa: F401
