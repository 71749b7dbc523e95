"""Fused SGD (momentum + weight decay) over the flat parameter buffer.

Same update as torch.optim.SGD as the reference configures it (main_embedding.py:385-388): weight decay is
added to the gradient before the momentum update; two parameter groups (backbone at 0.1*lr, classifier at
lr).  One streaming kernel per contiguous range instead of 338 per-tensor updates.
"""
from __future__ import annotations

import torch

from . import _lib


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.9, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._ranges = None
        self.grad_scale = 1.0

    @staticmethod
    def _store_of(p, stores):
        for st in stores:
            try:
                return st, st._index(p)
            except KeyError:
                continue
        return None, -1

    def bind(self, *models):
        """Tell the optimizer which models' flat stores its parameters live in."""
        self._stores = [m._engine.store for m in models]
        self._ranges = None
        return self

    def _build_ranges(self):
        ranges = []
        for gi, group in enumerate(self.param_groups):
            items = []
            for p in group["params"]:
                st, idx = self._store_of(p, self._stores)
                if st is None:
                    raise RuntimeError("FusedSGD: parameter does not belong to a bound model")
                n = (p.numel() + st.ALIGN - 1) // st.ALIGN * st.ALIGN
                items.append((id(st), st.offsets[idx], n, st))
            items.sort(key=lambda t: (t[0], t[1]))
            cur = None
            for sid, off, n, st in items:
                if cur is not None and cur[0] is st and cur[1] + cur[2] == off:
                    cur[2] += n
                else:
                    cur = [st, off, n]
                    ranges.append((gi, cur))
        self._ranges = ranges

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """torch's load_state_dict replaces every momentum buffer by a detached copy.  Before the first step that copy is
        carried into the flat buffer when it is created (step()); AFTER it -- a mid-run reload -- the kernel would keep
        updating the old flat buffer and the restored momentum would be silently ignored: copy the loaded buffers into the
        flat views and re-point the state at them."""
        super().load_state_dict(state_dict)
        for st in getattr(self, "_stores", ()):
            if st.flat_v is None:
                continue
            mine = {id(q) for g in self.param_groups for q in g["params"]}
            for p, off in zip(st.params, st.offsets):
                if id(p) not in mine:
                    continue
                view = st._view(st.flat_v, off, p)
                old = self.state[p].get("momentum_buffer") if p in self.state else None
                if old is None:
                    view.zero_()
                elif old.data_ptr() != view.data_ptr():
                    if tuple(old.shape) != tuple(p.shape):
                        raise RuntimeError("FusedSGD: restored momentum buffer of shape %s does not match its parameter %s"
                                           % (tuple(old.shape), tuple(p.shape)))
                    view.copy_(old.to(device=view.device, dtype=view.dtype))
                self.state[p]["momentum_buffer"] = view

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not hasattr(self, "_stores"):
            raise RuntimeError("call FusedSGD.bind(model) once before step()")
        lib = _lib.load()
        if self._ranges is None:
            self._build_ranges()
        for st in self._stores:
            if st.flat_p is None:
                return loss                       # no forward/backward happened yet
            if st.flat_v is None or st.flat_v.device != st.flat_p.device:
                st.flat_v = torch.zeros_like(st.flat_p)
                mine = {id(q) for g in self.param_groups for q in g["params"]}
                for p, off in zip(st.params, st.offsets):
                    if id(p) in mine:            # a parameter outside every group has no optimizer state (state_dict())
                        view = st._view(st.flat_v, off, p)
                        # momentum restored by load_state_dict() before the store was bound (--continue_training,
                        # main_embedding.py:421-434 of the reference), or left over from a re-bind: carry it into the
                        # flat buffer (OIHW -> the K-R-S-C view) instead of restarting from zero
                        old = self.state[p].get("momentum_buffer") if p in self.state else None
                        if old is not None and old.data_ptr() != view.data_ptr():
                            if tuple(old.shape) != tuple(p.shape):
                                raise RuntimeError("FusedSGD: restored momentum buffer of shape %s does not match its "
                                                   "parameter %s" % (tuple(old.shape), tuple(p.shape)))
                            view.copy_(old.to(device=view.device, dtype=view.dtype))
                        self.state[p]["momentum_buffer"] = view
            # gradients that are not views of the flat buffer (foreign autograd use) are folded in
            for p, gv in zip(st.params, st.grad_views):
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
        for gi, (st, off, n) in self._ranges:
            g = self.param_groups[gi]
            stream = torch.cuda.current_stream(st.flat_p.device).cuda_stream
            _lib.check(lib.dml_sgd_step(st.flat_p.data_ptr() + 4 * off, st.flat_g.data_ptr() + 4 * off,
                                        st.flat_v.data_ptr() + 4 * off, n, float(g["lr"]), float(g["momentum"]),
                                        float(g["weight_decay"]), float(self.grad_scale), stream), "dml_sgd_step")
        for st in self._stores:
            st.version += 1
        return loss
