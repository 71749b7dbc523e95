#!/usr/bin/env python3
"""Open-world evaluation step of the reference (test_embedding.py:225-653 there) on the MI355X path:
eval-mode forward, argmax, max-softmax score, dissum anomaly map and the novel-prototype relabel -- all on
the device (the reference copies a 16 x H x W logit tensor to the host per image).  Images are sharded
round-robin over ranks; the only collectives are the two sums at the very end (confusion matrix, per-image measures).

    python test_embedding.py --synthetic --height 1024 --width 2048 --num_images 4 [--ckpt X.pth]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import network  # noqa: E402
import utils  # noqa: E402
from dmlnet import parallel  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="deeplabv3plus_embedding_resnet101")
    p.add_argument("--num_classes", type=int, default=16)
    p.add_argument("--output_stride", type=int, default=16, choices=[8, 16])
    p.add_argument("--ckpt", default=None)
    p.add_argument("--prototype_json", default=None, help="k-shot prototype vectors (prototype_car_5_shot.json)")
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=2048)
    p.add_argument("--num_images", type=int, default=4)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16x2", "f32x3"],
                   help="bf16: bf16 storage (throughput mode); f32: exact fp32 MFMA (the reference's arithmetic); f16x2 / f32x3: fp32 tensors with the convolution products on the fp16 / bf16 matrix cores (fp32-accurate splits, bench.py's headline is f16x2)")
    o = p.parse_args()
    if not o.synthetic:
        raise SystemExit("only --synthetic data is available (datasets are outside the hot path)")
    rank, local, world = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1)              # without --ckpt every rank must still build the same (random-init) model
    model = getattr(network, o.model)(num_classes=o.num_classes, output_stride=o.output_stride,
                                      pretrained_backbone=False)
    if o.ckpt:
        model.load_state_dict(torch.load(o.ckpt, map_location="cpu")["model_state"])    # :748-749
    model.to(dev).eval()                                                                # :773
    model.set_compute_dtype(torch.bfloat16 if o.dtype == "bf16" else torch.float32,
                            fp32_products={"f32": "exact", "f32x3": "bf16x3", "f16x2": "f16x2"}.get(o.dtype))
    if o.prototype_json:
        proto = utils.mean_prototype(json.load(open(o.prototype_json)))                 # :245-258
    else:
        proto = np.full((o.num_classes,), 0.1)
    import anom_utils
    import metrics as metrics_mod
    seg_metrics = metrics_mod.StreamSegMetrics(o.num_classes + 1)                      # 16 known classes + the novel one
    aurocs, auprs, fprs = [], [], []
    n, t0 = 0, None
    with torch.no_grad():
        for i in range(rank, o.num_images, world):
            g = torch.Generator().manual_seed(4321 + i)
            img = torch.randn(1, 3, o.height, o.width, generator=g).to(dev)
            # synthetic ground truth: blocky train ids, one block-class plays the unknown object (label num_classes)
            coarse = torch.randint(0, o.num_classes + 1, (1, (o.height + 63) // 64, (o.width + 63) // 64), generator=g)
            target = coarse.repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :o.height, :o.width].contiguous().to(dev)
            outputs, centers, features = model(img)                                     # :337
            preds, msp = utils.argmax_msp(outputs)                                      # :339-342
            score = utils.dissum_score(outputs, clip=1000.0, inclusive=False)           # :349-350,365
            preds = utils.novel_relabel(preds, outputs, features, proto, -1.5, o.num_classes)   # :428-445
            seg_metrics.update(target, preds)                                           # :455 (metrics.update), on the device
            # pixel-level OOD measures of the anomaly score (eval_ood_traditional.py:128-148; the reference's `conf`
            # is a confidence, i.e. minus the anomaly score)
            res = anom_utils.eval_ood_measure(-score.reshape(-1).float(), target.reshape(-1), [o.num_classes])
            if res is not None:
                aurocs.append(res[0]); auprs.append(res[1]); fprs.append(res[2])
            if t0 is None:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            else:
                n += 1
    torch.cuda.synchronize()
    n_meas = len(aurocs)
    if world > 1:
        # every rank has scored its shard of the images: sum the confusion matrix and the per-image measures (sum,
        # count) over the ranks, so that rank 0 reports the whole evaluation set, as the reference's single process does
        import torch.distributed as dist
        seg_metrics.all_reduce()
        tot = torch.tensor([float(np.sum(aurocs)), float(np.sum(auprs)), float(np.sum(fprs)), float(n_meas)],
                           dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        n_meas = int(tot[3].item())
        mean_meas = (tot[:3] / max(n_meas, 1)).tolist()
    else:
        mean_meas = [float(np.mean(v)) if v else float("nan") for v in (aurocs, auprs, fprs)]
    if rank == 0:
        print(seg_metrics.to_str(seg_metrics.get_results()))
        if n_meas:          # frames without a novel-class pixel have no OOD measures (anom_utils.eval_ood_measure -> None)
            anom_utils.print_measures(mean_meas[0], mean_meas[1], mean_meas[2], "dissum")
    if n:
        print("rank %d: %.2f img/s at %dx%d (%d novel-class pixels in the last image, score mean %.4f)"
              % (rank, n / (time.perf_counter() - t0), o.height, o.width, int((preds == o.num_classes).sum()),
                 float(score.mean())))


if __name__ == "__main__":
    main()
