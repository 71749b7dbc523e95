"""Open-set evaluation driver of the anomaly sub-project on the MI355X path (reference: anomaly/eval_ood_traditional.py,
`evaluate` :150-560 and `main` :563-640, with config/test_ood_street.yaml: resnet50dilated + ppm_deepsup_embedding,
13 classes, imgSizes (300, 375, 450, 525, 600), imgMaxSize 1000, padding_constant 8).

Per frame: the resized copies go through `models.evaluate_multiscale` (the loop of :198-210, mean folded into the
upsample kernel, scales concurrent), `pred = argmax` (:218), the confidence map of `--ood`
(:275-340: msp | maxlogit | dissum | background; the CRF / kNN variants of the reference need pydensecrf / are
plotting experiments and are not offered), then `eval_ood_measure` (:128-148) and the pixel accuracy / IoU meters
(:548-556) -- all on the device; nothing but the three OOD numbers and the confusion counts per frame reaches the host.
Data: `--synthetic` frames only (decoding / PIL-resizing StreetHazards PNGs is dataset IO, outside the build's scope).
"""
import argparse
import time

import numpy as np
import torch

import anom_utils
import metrics as metrics_mod
import models
import utils

IMG_SIZES, IMG_MAX_SIZE, PADDING_CONSTANT = (300, 375, 450, 525, 600), 1000, 8


def resized_shapes(h, w):
    """dataset.py's TestDataset sizes: short side -> each of IMG_SIZES, long side <= IMG_MAX_SIZE, both rounded up to a
    multiple of PADDING_CONSTANT."""
    out = []
    for short in IMG_SIZES:
        sc = min(short / float(min(h, w)), IMG_MAX_SIZE / float(max(h, w)))
        th, tw = int(h * sc), int(w * sc)
        out.append(((th + PADDING_CONSTANT - 1) // PADDING_CONSTANT * PADDING_CONSTANT,
                    (tw + PADDING_CONSTANT - 1) // PADDING_CONSTANT * PADDING_CONSTANT))
    return out


def confidence(scores, ood, exclude_back=False):
    """:275-340.  scores [1, K, H, W] on the device -> conf [H, W] on the device."""
    tmp = scores[:, 1:].contiguous() if exclude_back else scores
    if ood == "msp":
        return utils.argmax_msp(tmp)[1][0]
    if ood == "maxlogit":
        preds = utils.argmax_msp(tmp)[0]
        return tmp.gather(1, preds.unsqueeze(1))[0, 0]
    if ood == "dissum":
        return utils.dissum_score(tmp, clip=400.0, inclusive=True)[0]
    if ood == "background":
        return tmp[0, 0]
    raise NotImplementedError("--ood %s" % ood)


def evaluate(segmentation_module, frames, num_class, ood, out_labels, exclude_back=False):
    """frames: iterable of (img_resized_list, seg_label int64 [H, W]) on the device."""
    seg_metrics = metrics_mod.StreamSegMetrics(num_class)
    aurocs, auprs, fprs, times = [], [], [], []
    for imgs, seg_label in frames:
        torch.cuda.synchronize()
        tic = time.perf_counter()
        seg_size = tuple(seg_label.shape)
        scores, ft1 = models.evaluate_multiscale(segmentation_module, imgs, seg_size)
        pred = utils.argmax_msp(scores)[0]
        conf = confidence(scores, ood, exclude_back)
        res = anom_utils.eval_ood_measure(conf, seg_label, out_labels)
        if res is not None:
            aurocs.append(res[0]); auprs.append(res[1]); fprs.append(res[2])
        seg_metrics.update(seg_label[None], pred)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - tic)
    return {"auroc": float(np.mean(aurocs)) if aurocs else float("nan"),
            "aupr": float(np.mean(auprs)) if auprs else float("nan"),
            "fpr": float(np.mean(fprs)) if fprs else float("nan"),
            "seg": seg_metrics.get_results(), "sec_per_frame": float(np.mean(times[1:] or times))}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--ood", default="dissum", choices=["msp", "maxlogit", "dissum", "background"])
    p.add_argument("--exclude_back", action="store_true")
    p.add_argument("--out_label", type=int, default=13, help="cfg.OOD.out_labels: the anomaly id of StreetHazards")
    p.add_argument("--num_images", type=int, default=4)
    p.add_argument("--height", type=int, default=720)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--encoder_weights", default="")
    p.add_argument("--decoder_weights", default="")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16x2", "f32x3"],
                   help="bf16: bf16 storage (throughput mode); f32: exact fp32 MFMA (the reference's arithmetic); f16x2 / f32x3: fp32 tensors with the convolution products on the fp16 / bf16 matrix cores (fp32-accurate splits, bench.py's headline is f16x2)")
    p.add_argument("--synthetic", action="store_true")
    opts = p.parse_args()
    if not opts.synthetic:
        raise SystemExit("only --synthetic frames are available (dataset decoding is outside the hot path)")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    torch.manual_seed(304)
    enc = models.ModelBuilder.build_encoder("resnet50dilated", fc_dim=2048, weights=opts.encoder_weights)
    dec = models.ModelBuilder.build_decoder("ppm_deepsup_embedding", fc_dim=2048, num_class=13,
                                            weights=opts.decoder_weights, use_softmax=True)
    seg = models.SegmentationModuleOOD(enc, dec, None).to(device).eval()
    seg.set_compute_dtype(torch.bfloat16 if opts.dtype == "bf16" else torch.float32,
                            fp32_products={"f32": "exact", "f32x3": "bf16x3", "f16x2": "f16x2"}.get(opts.dtype))
    g = torch.Generator().manual_seed(7)
    shapes = resized_shapes(opts.height, opts.width)

    def frames():
        for _ in range(opts.num_images):
            imgs = [torch.randn(1, 3, h, w, generator=g).to(device) for h, w in shapes]
            coarse = torch.randint(0, 14, ((opts.height + 31) // 32, (opts.width + 31) // 32), generator=g)
            lab = coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:opts.height, :opts.width].contiguous()
            yield imgs, lab.to(device)

    r = evaluate(seg, frames(), 14, opts.ood, (opts.out_label,), opts.exclude_back)
    print("mean auroc = ", r["auroc"], "mean aupr = ", r["aupr"], " mean fpr = ", r["fpr"])          # :587-589
    print("Mean IoU: %.4f, Accuracy: %.2f%%, Inference Time: %.4fs" % (r["seg"]["Mean IoU"], 100.0 * r["seg"]["Overall Acc"],
                                                                        r["sec_per_frame"]))


if __name__ == "__main__":
    main()
