"""Measurement (GPU box): is the bf16 plan's gradient noise the bf16-storage emulation's?  One draw of the end-to-end
comparison (tests/test_gpu_bf16_parity.py) is itself a random variable -- the ASPP image-pooling BatchNorm over B samples
amplifies whatever perturbation reaches it, so the per-tensor 1 - cos of BOTH implementations moves by +-20 % from input
to input -- hence this tool: S seeds (weights and inputs), per seed the median / p95 over the 338 parameter tensors of
1 - cos to the fp32 oracle for (a) the HIP plan with fp32 staging of multi-producer gradients, (b) the same plan rounding
after every producer (DML_GRAD_STAGE32=0), (c) the emulated bf16-storage oracle; then mean and standard error over seeds.
usage: python tests/tools/bf16_noise_seeds.py 4x3x256x256 9,10,11,12,13,14,15,16"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import numpy as np, torch
import helpers as H
import test_gpu_bf16_parity as P
import utils

shape = tuple(int(v) for v in sys.argv[1].split("x"))
seeds = [int(v) for v in sys.argv[2].split(",")]
torch.set_num_threads(min(64, torch.get_num_threads() or 8))


def hip_grads(seed, tag, stage):
    os.environ["DML_GRAD_STAGE32"] = "1" if stage else "0"
    img = H.synth_tensor(seed, tag + ".img", shape)
    lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
    m = P._build_hip(torch.bfloat16, seed)
    lg, _, ft = m(img.cuda())
    utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
    del m
    torch.cuda.empty_cache()
    return g


rows = []
for seed in seeds:
    tag = "noise.%d" % seed
    _, _, g_true, _ = P._oracle_grads(seed, shape, tag, emulate=False)
    _, _, g_emu, _ = P._oracle_grads(seed, shape, tag, emulate=True)
    res = {}
    for name, g in (("staged", hip_grads(seed, tag, True)), ("per-producer", hip_grads(seed, tag, False)), ("emulation", g_emu)):
        c, r = P._vs_truth(g, g_true)
        res[name] = (np.median(c), np.percentile(c, 95), c.max(), np.median(r))
    rows.append(res)
    print("seed %3d | " % seed + " | ".join("%s: median %.3e p95 %.3e max %.3e norm %.3f" % ((k,) + v) for k, v in res.items()), flush=True)
print("\n%d seeds, %s: mean +- standard error over seeds" % (len(seeds), "x".join(map(str, shape))))
for k in rows[0]:
    a = np.array([r[k] for r in rows])
    print("%-14s median 1-cos %.4f +- %.4f | p95 %.4f +- %.4f | max %.4f +- %.4f | norm ratio %.3f"
          % (k, a[:, 0].mean(), a[:, 0].std(ddof=1) / np.sqrt(len(a)), a[:, 1].mean(), a[:, 1].std(ddof=1) / np.sqrt(len(a)),
             a[:, 2].mean(), a[:, 2].std(ddof=1) / np.sqrt(len(a)), a[:, 3].mean()))
e = np.array([r["emulation"] for r in rows])
for k in ("staged", "per-producer"):
    a = np.array([r[k] for r in rows])
    q = a[:, :3] / e[:, :3]
    print("%-14s / emulation, per seed then averaged: median x%.3f +- %.3f | p95 x%.3f +- %.3f | max x%.3f +- %.3f"
          % (k, q[:, 0].mean(), q[:, 0].std(ddof=1) / np.sqrt(len(q)), q[:, 1].mean(), q[:, 1].std(ddof=1) / np.sqrt(len(q)),
             q[:, 2].mean(), q[:, 2].std(ddof=1) / np.sqrt(len(q))))
