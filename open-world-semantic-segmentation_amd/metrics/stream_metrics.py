"""Streaming segmentation metrics of the reference (metrics/stream_metrics.py there) with the confusion matrix kept
on the MI355X: `update` takes the label / prediction tensors where the model left them (CUDA int64) instead of
`labels.cpu().numpy()` / `preds` copies of every batch (main_embedding.py:267-269 of the reference), and only the
n x n matrix comes to the host in `get_results`.  Same class, method and result-key names.  No CPU fallback.
"""
import numpy as np
import torch

from dmlnet import _lib


class StreamSegMetrics(object):
    def __init__(self, n_classes):
        # the reference sets self.n_classes = 19 whatever it is given (stream_metrics.py:29) and therefore only works
        # for 19 classes; here the argument is honoured (identical for 19)
        self.n_classes = int(n_classes)
        self.confusion_matrix = None          # device int64 [n, n], created on the first update
        self._host = np.zeros((self.n_classes, self.n_classes))

    def update(self, label_trues, label_preds):
        lib = _lib.load()
        for t in (label_trues, label_preds):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64):
                raise TypeError("StreamSegMetrics.update takes int64 CUDA tensors (there is no CPU fallback)")
        if label_trues.shape != label_preds.shape:
            raise ValueError("label_trues and label_preds differ in shape")
        lt, lp = label_trues.contiguous(), label_preds.contiguous()
        if lt.data_ptr() & 15:
            lt = lt.clone()                   # the kernel reads pairs of int64 with 16-byte loads
        if lp.data_ptr() & 15:
            lp = lp.clone()
        if self.confusion_matrix is None or self.confusion_matrix.device != lt.device:
            self.confusion_matrix = torch.zeros((self.n_classes, self.n_classes), dtype=torch.int64, device=lt.device)
        st = torch.cuda.current_stream(lt.device).cuda_stream
        _lib.check(lib.dml_confusion_update(lt.data_ptr(), lp.data_ptr(), self.confusion_matrix.data_ptr(), lt.numel(),
                                            self.n_classes, st), "dml_confusion_update")

    @staticmethod
    def to_str(results):
        """one "name: value" line per scalar score; the per-class table is left out (stream_metrics.py:38-47)"""
        lines = ["%s: %f" % (name, value) for name, value in results.items() if name != "Class IoU"]
        return "\n" + "\n".join(lines) + "\n"

    def all_reduce(self, group=None):
        """Sum the confusion matrix over the ranks of a data-parallel evaluation (each rank has seen its shard of the
        images); afterwards every rank reports the scores of the whole set.  No-op outside a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        if self.confusion_matrix is None:
            # a rank that scored no image (fewer images than ranks) must still issue the SAME collective as the others:
            # skipping it would pair this rank's next all_reduce with their matrix all_reduce (hang / corrupted sums)
            self.confusion_matrix = torch.zeros((self.n_classes, self.n_classes), dtype=torch.int64,
                                                device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(self.confusion_matrix, group=group)

    def get_results(self):
        """The scores of stream_metrics.py:57-83 of the reference from the n x n count matrix (rows = true class,
        columns = prediction), in float64 like there: pixel accuracy, mean per-class accuracy, mean IoU and the
        frequency-weighted IoU (classes that never occur give NaN rows and are skipped by the nan-means)."""
        if self.confusion_matrix is None:
            counts = self._host
        else:
            counts = self.confusion_matrix.cpu().numpy().astype(np.float64)
        tp = np.diag(counts)
        per_true, per_pred, total = counts.sum(axis=1), counts.sum(axis=0), counts.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = tp / (per_true + per_pred - tp)
            scores = {
                "Overall Acc": tp.sum() / total,
                "Mean Acc": np.nanmean(tp / per_true),
                "FreqW Acc": ((per_true / total)[per_true > 0] * iou[per_true > 0]).sum(),
                "Mean IoU": np.nanmean(iou),
            }
        scores["Class IoU"] = {c: iou[c] for c in range(self.n_classes)}
        return scores

    def reset(self):
        if self.confusion_matrix is not None:
            self.confusion_matrix.zero_()


class AverageMeter(object):
    """Named running means -- the host-side helper the reference exports next to StreamSegMetrics
    (metrics/stream_metrics.py:86-114 there): update(id, val) adds a sample, get_results(id) is the mean so far,
    reset(id) zeroes one entry and reset_all() forgets every name.  `book` maps id -> [sum, count] as there."""

    def __init__(self):
        self.book = {}

    def update(self, id, val):
        entry = self.book.setdefault(id, [0, 0])
        entry[0] = entry[0] + val if entry[1] else val
        entry[1] += 1

    def get_results(self, id):
        if id not in self.book:
            raise AssertionError("no samples recorded under %r" % (id,))
        total, count = self.book[id]
        return total / count

    def reset(self, id):
        if id in self.book:
            self.book[id][:] = [0, 0]

    def reset_all(self):
        self.book = {}
