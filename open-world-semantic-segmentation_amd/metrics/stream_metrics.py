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
        string = "\n"
        for k, v in results.items():
            if k != "Class IoU":
                string += "%s: %f\n" % (k, v)
        return string

    def get_results(self):
        """overall accuracy, mean accuracy, mean IoU, frequency-weighted accuracy (stream_metrics.py:57-83)"""
        hist = self._host if self.confusion_matrix is None else self.confusion_matrix.cpu().numpy().astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(hist).sum() / hist.sum()
            acc_cls = np.diag(hist) / hist.sum(axis=1)
            acc_cls = np.nanmean(acc_cls)
            iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
            mean_iu = np.nanmean(iu)
            freq = hist.sum(axis=1) / hist.sum()
            fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        cls_iu = dict(zip(range(self.n_classes), iu))
        return {"Overall Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu, "Class IoU": cls_iu}

    def reset(self):
        if self.confusion_matrix is not None:
            self.confusion_matrix.zero_()


class AverageMeter(object):
    """Computes average values (stream_metrics.py:88-116 of the reference; host-side bookkeeping)"""

    def __init__(self):
        self.book = dict()

    def reset_all(self):
        self.book.clear()

    def reset(self, id):
        item = self.book.get(id, None)
        if item is not None:
            item[0] = 0
            item[1] = 0

    def update(self, id, val):
        record = self.book.get(id, None)
        if record is None:
            self.book[id] = [val, 1]
        else:
            record[0] += val
            record[1] += 1

    def get_results(self, id):
        record = self.book.get(id, None)
        assert record is not None
        return record[0] / record[1]
