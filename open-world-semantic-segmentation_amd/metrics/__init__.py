from .stream_metrics import StreamSegMetrics  # noqa: F401
