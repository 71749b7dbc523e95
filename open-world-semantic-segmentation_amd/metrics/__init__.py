from .stream_metrics import StreamSegMetrics, AverageMeter  # noqa: F401
