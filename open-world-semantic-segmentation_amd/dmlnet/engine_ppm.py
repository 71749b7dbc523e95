"""Inference plan of the anomaly sub-project's segmentation model: dilated deep-stem ResNet encoder + pyramid-pooling
embedding decoder (anomaly/models/models.py:285-346,586-687 and anomaly/models/resnet.py:96-166 of the reference;
SURVEY.md 8(f) rank 2).  Built from the same plan pieces as the DeepLab model (`Plan.cbr`: conv with BatchNorm running
statistics + residual + ReLU in its epilogue), plus three kernels of its own: adaptive average pooling on NHWC, the
13-prototype distance at 1/8 resolution, and the bilinear upsample into the NCHW score tensor with the multi-scale mean
of eval_ood_traditional.py:198-210 folded into the store.  Training with this decoder is not offered: the reference's own
training path for it is dead.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib
from .engine import _PAD_CIN, Engine, Plan, _round_up


class PPMPlan(Plan):
    def build(self):
        lib = self.lib
        if self.training:
            raise NotImplementedError("the pyramid-pooling embedding decoder is inference-only on the MI355X path "
                                      "(the reference cannot train it either: SURVEY.md F11)")
        B, H, W = self.B, self.H, self.W
        enc, dec = self.e.model.backbone, self.e.model.decoder
        for mod in list(enc.modules()) + list(dec.modules()):
            if isinstance(mod, nn.BatchNorm2d) and mod.training:
                raise NotImplementedError("a BatchNorm2d in train() mode inside an eval() model is not supported")
        self.pre_prep, self.bn_eval = [], []
        self.bn_eval_args = self.call(self.fwd, lib.dml_bn_eval_coeffs_table, 0, 0)
        x_in = self.new(B, H, W, _PAD_CIN)
        self.images_args = self.call(self.fwd, lib.dml_pack_input, 0, x_in.ptr, B, 3, H, W, _PAD_CIN, self.dt)

        # deep stem: three 3x3 convs (the first with stride 2) + 3x3 s2 max pool (resnet.py:100-110,155-159)
        s1 = self.cbr(x_in, enc.conv1, enc.bn1, need_dgrad=False)
        s2 = self.cbr(s1.z, enc.conv2, enc.bn2)
        s3 = self.cbr(s2.z, enc.conv3, enc.bn3)
        z0 = s3.z
        Hp, Wp = (z0.H - 1) // 2 + 1, (z0.W - 1) // 2 + 1
        p0 = self.new(B, Hp, Wp, z0.C)
        self.call(self.fwd, lib.dml_maxpool3x3s2_fwd, z0.ptr, p0.ptr, None, B, z0.H, z0.W, z0.C, self.dt)

        # bottlenecks (resnet.py:73-93; strides already turned into dilations by ResnetDilated, models.py:290-328).
        # The last block writes conv5 straight into the first 2048 channels of the pyramid concat.
        blocks = [blk for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4) for blk in layer]
        n_ppm = len(dec.ppm)
        fc_dim = blocks[-1].conv3.out_channels
        ppm_c = dec.ppm[0][1].out_channels if n_ppm else 0
        x, cat = p0, None
        for bi, blk in enumerate(blocks):
            u1 = self.cbr(x, blk.conv1, blk.bn1)
            u2 = self.cbr(u1.z, blk.conv2, blk.bn2)
            idt = x
            if blk.downsample is not None:
                idt = self.cbr(x, blk.downsample[0], blk.downsample[1], relu=False).z
            out = None
            if bi == len(blocks) - 1:
                cat = self.new(B, u2.z.H, u2.z.W, fc_dim + n_ppm * ppm_c)
                out = cat.slice(0, fc_dim)
            x = self.cbr(u2.z, blk.conv3, blk.bn3, relu=True, res=idt, out=out).z
        conv5, h, w = x, x.H, x.W

        # pyramid pooling (models.py:620-631): AdaptiveAvgPool2d(s) -> 1x1 conv -> BN -> ReLU -> bilinear back to h x w
        for i, branch in enumerate(dec.ppm):
            S = branch[0].output_size
            S = int(S[0] if isinstance(S, (tuple, list)) else S)
            if S > h or S > w:
                raise ValueError("input too small for the %dx%d pooling bin grid (features are %dx%d)" % (S, S, h, w))
            pooled = self.new(B, S, S, fc_dim)
            ws = self.fbuf(int(lib.dml_adaptive_avgpool_ws_elems(B, h, w, fc_dim, S)))
            self.call(self.fwd, lib.dml_adaptive_avgpool_fwd, conv5.ptr, pooled.ptr, ws.data_ptr(), B, h, w, fc_dim, conv5.ld,
                      S, self.dt)
            u = self.cbr(pooled, branch[1], branch[2])
            self.call(self.fwd, lib.dml_bilinear_fwd, u.z.ptr, cat.slice(fc_dim + ppm_c * i, ppm_c).ptr, B, S, S, h, w, ppm_c,
                      u.z.ld, cat.ld, self.dt, 0, 0)

        # conv_last (models.py:602-609): 3x3 -> BN -> ReLU -> Dropout2d (identity in eval) -> 1x1 (+bias) = the embedding
        ulast = self.cbr(cat, dec.conv_last[0], dec.conv_last[1])
        fin = dec.conv_last[4]
        K = fin.out_channels
        Kp = _round_up(K, 8)
        if K != 13:
            # the reference hard-codes 13 centers (models.py:613-617); any other class count fails in its subtraction
            raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (13) at non-singleton dimension 3" % K)
        cin = fin.in_channels
        wpad, bpad = self.fbuf(Kp * cin, zero=True), self.fbuf(Kp, zero=True)

        def stage(wpad=wpad, bpad=bpad, fin=fin, K=K, cin=cin):
            wpad[:K * cin].copy_(fin.weight.detach().reshape(-1))
            if fin.bias is not None:
                bpad[:K].copy_(fin.bias.detach())
        self.pre_prep.append(stage)
        w_fin, _ = self.prep_weight(fin, cin, False, src_ptr=wpad.data_ptr(), N=Kp)
        emb = self.new(B, h, w, Kp, f32=True)
        self.conv_fwd(ulast.z, fin, emb, w_fin, None, bias_ptr=bpad.data_ptr() if fin.bias is not None else None, N=Kp)
        # distance to the 13 prototypes at 1/8 resolution (models.py:633-657), THEN the upsample to segSize (:659-668)
        protos = self.e.prototypes_padded(K, Kp)
        dist = self.new(B, h, w, Kp, f32=True)
        self.call(self.fwd, lib.dml_proto_dist_nhwc, emb.ptr, protos.data_ptr(), dist.ptr, B * h * w, K, Kp, Kp, Kp)
        self.up_scores = self.call(self.fwd, lib.dml_upsample_nhwc_to_nchw, dist.ptr, 0, B, h, w, Kp, K, 1, 1, 1.0, 0)
        self.up_feats = self.call(self.fwd, lib.dml_upsample_nhwc_to_nchw, emb.ptr, 0, B, h, w, Kp, K, 1, 1, 1.0, 0)
        self.K, self.Kp, self.heads = K, Kp, []
        self.nbt_inc = None
        arr = (_lib.BnEvalDesc * len(self.bn_eval))(*self.bn_eval)
        self.bn_eval_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.bn_eval_args[0], self.bn_eval_args[1] = self.bn_eval_table.data_ptr(), len(self.bn_eval)


class PPMEngine(Engine):
    plan_cls = PPMPlan

    def _prepare(self, x, seg_size, dtype, scores, feats, alpha, accumulate):
        if not x.is_cuda:
            raise RuntimeError("the MI355X path needs a ROCm device tensor (got %s); there is no CPU fallback" % x.device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected input [B,3,H,W], got %s" % (tuple(x.shape),))
        x = x.contiguous().float()
        plan = self.plan_for(x, dtype, False)
        plan.refresh_weights(torch.cuda.current_stream(x.device).cuda_stream)
        B = x.shape[0]
        Hs, Ws = int(seg_size[0]), int(seg_size[1])
        for t in (scores, feats):
            if tuple(t.shape) != (B, plan.K, Hs, Ws) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device:
                raise ValueError("accumulators must be contiguous float32 [B, %d, %d, %d] on the input's device" % (plan.K, Hs, Ws))
        plan.images_args[0] = x.data_ptr()
        for args, dst in ((plan.up_scores, scores), (plan.up_feats, feats)):
            args[1], args[7], args[8], args[9], args[10] = dst.data_ptr(), Hs, Ws, float(alpha), 1 if accumulate else 0
        plan.last_input = x
        return plan

    def infer(self, x: torch.Tensor, seg_size, dtype: torch.dtype, scores=None, feats=None, alpha: float = 1.0):
        """One forward at x's resolution; scores / feats [B, 13, *seg_size] (+)= alpha * upsampled result."""
        accumulate = scores is not None
        if scores is None:
            shape = (x.shape[0], 13, int(seg_size[0]), int(seg_size[1]))
            scores = torch.empty(shape, dtype=torch.float32, device=x.device)
            feats = torch.empty(shape, dtype=torch.float32, device=x.device)
        plan = self._prepare(x, seg_size, dtype, scores, feats, alpha, accumulate)
        Plan.run(plan.fwd, torch.cuda.current_stream(x.device).cuda_stream)
        return scores, feats

    use_graphs = os.environ.get("DML_PPM_GRAPH", "1") == "1"

    def _replay_body(self, plan, s):
        """The body of a scale (everything but the two upsample-accumulate launches) as one hipGraph: ~190 launches of a
        few microseconds each are host-bound when five scales run side by side.  The input is copied into a buffer the
        graph owns; weights are prepared outside (refresh_weights), so a replay is valid until they change."""
        x = plan.last_input
        key = plan.prepped_version
        if getattr(plan, "_graph", None) is None or plan._graph_key != key:
            plan._static_in = torch.empty_like(x)
            plan.images_args[0] = plan._static_in.data_ptr()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                plan._static_in.copy_(x)
                Plan.run(plan.fwd, s.cuda_stream, 0, len(plan.fwd) - 2)        # warm-up outside the capture
                s.synchronize()
                with torch.cuda.graph(g, stream=s):
                    Plan.run(plan.fwd, s.cuda_stream, 0, len(plan.fwd) - 2)
            plan._graph, plan._graph_key = g, key
        with torch.cuda.stream(s):
            plan._static_in.copy_(x)
            plan._graph.replay()

    def infer_multiscale(self, imgs, seg_size, dtype: torch.dtype):
        """eval_ood_traditional.py:190-210 for the resized copies of one frame: scores = sum_i pred_i / n (same for the
        features).  The copies are small (grids of 40-200 workgroups), so their plans run CONCURRENTLY, one HIP stream
        each; only the two upsample-accumulate launches of every scale are ordered, on the caller's stream."""
        n = len(imgs)
        dev = imgs[0].device
        main = torch.cuda.current_stream(dev)
        shape = (imgs[0].shape[0], 13, int(seg_size[0]), int(seg_size[1]))
        scores = torch.empty(shape, dtype=torch.float32, device=dev)
        feats = torch.empty(shape, dtype=torch.float32, device=dev)
        if not hasattr(self, "_scale_streams"):
            self._scale_streams = {}
        pool = self._scale_streams.setdefault(dev, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=dev))
        first, todo = True, list(range(n))
        while todo:
            wave, later, seen = [], [], set()
            for i in todo:                      # one use of a plan (= input shape) per wave: its buffers are not re-entrant
                if tuple(imgs[i].shape) in seen:
                    later.append(i)
                    continue
                seen.add(tuple(imgs[i].shape))
                plan = self._prepare(imgs[i], seg_size, dtype, scores, feats, 1.0 / n, True)
                # snapshot the per-call arguments of the tail (a later wave re-patches the same lists)
                tail = [(fn, list(args)) for fn, args in plan.fwd[-2:]]
                wave.append((plan, tail, pool[len(wave)]))
            for plan, tail, s in wave:
                s.wait_stream(main)
                if self.use_graphs:
                    self._replay_body(plan, s)
                else:
                    Plan.run(plan.fwd, s.cuda_stream, 0, len(plan.fwd) - 2)
            for plan, tail, s in wave:
                main.wait_stream(s)
                for fn, args in tail:
                    args[10] = 0 if first else 1
                    rc = fn(*args, main.cuda_stream)
                    if rc:
                        _lib.check(rc, "upsample-accumulate")
                first = False
            todo = later
        return scores, feats
