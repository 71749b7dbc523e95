"""Incremental-head training driver: the recipe of the reference's main_self_distillation.py (:330-507) on the MI355X
path.  A trained 16-class DMLNet (`--ckpt`, keys that exist in the two-head model are taken over, :397-404) gets a
second, 17-prototype head `classifier_1`; only that head is in the optimizer (:354-357), every BatchNorm2d runs on its
running statistics (:432-435), the loss reads the LAST head's output (:497-499) and the first head only produces
pseudo-labels (:447-451).  The reference file is a research script caught mid-edit (plots every batch, several
mutually exclusive label remaps commented in and out); what is reproduced here is the recipe those variants share:

  * labels: held-out class (train id 13, car) -> new id 16, ids above move down -- the "with all annotations" remap
    (:470-473) = the evaluation remap of test_embedding.py:448-451, composed into the crop kernel's label table;
  * `--pseudo_labels`: ignored pixels take the base head's prediction, overridden by a later head where that head
    predicts its own novel class (:447-451 + the commented `labels[labels == 255] = labels_base[labels == 255]`).

Trunk and base head have requires_grad = False, so the backward plan stops at the new head's inputs (the reference
computes and discards those gradients).  Data: `--synthetic` only, as main_embedding.py.
"""
import argparse
import os
import time

import numpy as np
import torch

import network
import utils
from datasets import Cityscapes
from dmlnet import parallel
from dmlnet.optim import FusedSGD


def get_argparser():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="deeplabv3plus_embedding_self_distillation_resnet101")
    p.add_argument("--num_classes", type=int, default=16)
    p.add_argument("--novel_cls", type=int, default=1)              # :447: number of incremental heads
    p.add_argument("--output_stride", type=int, default=16, choices=[8, 16])
    p.add_argument("--total_itrs", type=int, default=1000)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--batch_size", type=int, default=16, help="GLOBAL batch size, sharded over the ranks")
    p.add_argument("--crop_size", type=int, default=768)
    p.add_argument("--ckpt", default=None, help="checkpoint of the base model (or of this model)")
    p.add_argument("--pseudo_labels", action="store_true")
    p.add_argument("--train_backbone", action="store_true", help="the commented optimizer of :348-351")
    p.add_argument("--save_dir", default="checkpoints")
    p.add_argument("--save_interval", type=int, default=0)
    p.add_argument("--print_interval", type=int, default=10)
    p.add_argument("--random_seed", type=int, default=1)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--frame_height", type=int, default=1024)
    p.add_argument("--frame_width", type=int, default=2048)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16x2", "f32x3"],
                   help="bf16: bf16 storage (throughput mode); f32: exact fp32 MFMA (the reference's arithmetic); f16x2 / f32x3: fp32 tensors with the convolution products on the fp16 / bf16 matrix cores (fp32-accurate splits, bench.py's headline is f16x2)")
    return p


def pseudo_labels(outputs, base_classes):
    """labels_base of :447-451: argmax of the base head; head i+1 overrides where it predicts ITS novel class."""
    base, _ = utils.argmax_msp(outputs[0])
    for i in range(1, len(outputs)):
        nxt, _ = utils.argmax_msp(outputs[i])
        novel = base_classes + i - 1
        base = torch.where(nxt == novel, nxt, base)
    return base


def main():
    opts = get_argparser().parse_args()
    if not opts.synthetic:
        raise SystemExit("only --synthetic data is available (datasets are outside the hot path)")
    if opts.novel_cls != 1 or opts.num_classes != 16:
        raise SystemExit("the two-head model of the reference is 16 + 17 prototypes (network/modeling.py:150-158)")
    rank, local, world = parallel.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(opts.random_seed)

    model = getattr(network, opts.model)(num_classes=opts.num_classes, output_stride=opts.output_stride,
                                         pretrained_backbone=False)
    utils.set_bn_momentum(model.backbone, momentum=0.01)                           # :345
    model.set_compute_dtype(torch.bfloat16 if opts.dtype == "bf16" else torch.float32,
                            fp32_products={"f32": "exact", "f32x3": "bf16x3", "f16x2": "f16x2"}.get(opts.dtype))
    groups = [{"params": model.classifier_1.parameters(), "lr": opts.lr}]          # :354-357
    if opts.train_backbone:
        groups.insert(0, {"params": model.backbone.parameters(), "lr": 0.1 * opts.lr})
    else:
        for p in list(model.backbone.parameters()) + list(model.classifier.parameters()):
            p.requires_grad_(False)
    optimizer = FusedSGD(groups, lr=opts.lr, momentum=0.9, weight_decay=opts.weight_decay).bind(model)
    scheduler = utils.PolyLR(optimizer, opts.total_itrs, power=0.9)
    criterion = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0, sync=True if world > 1 else None)  # :367

    cur_itrs = 0
    if opts.ckpt and os.path.isfile(opts.ckpt):                                    # :393-404: keys both models have
        ck = torch.load(opts.ckpt, map_location="cpu")
        own = model.state_dict()
        own.update({k: v for k, v in ck["model_state"].items() if k in own})
        model.load_state_dict(own)
    model.to(device)
    if world > 1:
        model._engine.store.bind(device)
        model._engine.reducer = parallel.GradReducer(model._engine.store, bucket_mb=32.0, average=False)

    from utils import ext_transforms as et
    lo, hi = parallel.shard_range(opts.batch_size, rank, world)
    g = torch.Generator().manual_seed(1234 + rank)
    fh, fw = max(opts.crop_size, opts.frame_height), max(opts.crop_size, opts.frame_width)
    frames = torch.randint(0, 256, (hi - lo, fh, fw, 3), generator=g, dtype=torch.uint8).to(device)
    coarse = torch.randint(0, 34, (hi - lo, (fh + 63) // 64, (fw + 63) // 64), generator=g, dtype=torch.uint8)
    frame_labels = coarse.repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :fh, :fw].contiguous().to(device)
    # raw id -> 17-id space of the shipped dataset (truck, bus unknown) -> car (13) becomes the novel class 16
    lut, lut_true = Cityscapes.label_luts([14, 15])
    lut = Cityscapes.eval_relabel_lut(held_out=13, new_id=16)[lut]
    transform = et.ExtCompose([
        et.ExtRandomCrop(size=(opts.crop_size, opts.crop_size)),
        et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
        et.ExtRandomHorizontalFlip(),
        et.ExtToTensor(),
        et.ExtNormalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
    ], label_luts=(lut, lut_true))

    interval_loss, t0 = None, time.perf_counter()
    while cur_itrs < opts.total_itrs:
        model.train()                                                              # :432-435
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        cur_itrs += 1
        images, labels, _ = transform(frames, frame_labels)
        optimizer.zero_grad()
        outputs, centers, features = model(images)                                 # lists, one entry per head
        if opts.pseudo_labels:
            with torch.no_grad():
                base = pseudo_labels([o.detach() for o in outputs], opts.num_classes)
                labels = torch.where(labels == 255, base, labels)
        loss = criterion(outputs[-1], labels, features[-1])                        # :497-499
        loss.backward()
        optimizer.step()
        scheduler.step()
        interval_loss = loss.detach() if interval_loss is None else interval_loss + loss.detach()
        if cur_itrs % opts.print_interval == 0:
            mean_loss = float(interval_loss) / opts.print_interval       # the D2H copy waits for the interval's kernels
            dt = time.perf_counter() - t0
            if rank == 0:
                print("Itrs %d/%d, Loss=%f, %.1f img/s" % (cur_itrs, opts.total_itrs, mean_loss,
                                                           opts.batch_size * opts.print_interval / dt))
            interval_loss, t0 = None, time.perf_counter()
        if opts.save_interval and cur_itrs % opts.save_interval == 0 and rank == 0:
            utils.mkdir(opts.save_dir)
            torch.save({"cur_itrs": cur_itrs, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(),
                        "scheduler_state": scheduler.state_dict(), "best_score": 0.0},
                       os.path.join(opts.save_dir, "latest_%s_synthetic.pth" % opts.model))


if __name__ == "__main__":
    main()
