#!/usr/bin/env python3
"""Mint the conditioning record of the NON-SQUARE live-oracle check (authoring container only; imports the real reference):

    g13n_nonsquare   deeplabv3plus_embedding_resnet101, 2 x 3 x 128 x 192, synth weights of seed 11 with the BatchNorm betas moved so
                     that no ReLU input of the network lies within 64 * eps32 * sum|terms| (and 6 x the reference's own fp32-vs-fp64
                     noise) of zero -- the procedure and the proof of tests/tools/mint_golden_large.py, on another input shape.

tests/test_gpu_model.py::test_against_oracle_nonsquare_strict runs the HIP model AND the oracle (fp32 and fp64, at test time) on these
weights; the fixture carries only the moved betas (sparse) and the proof numbers.  The 64 x 96 input that test used before normalises
layer3 / layer4 / ASPP over 48 samples and sat on ReLU knife edges: its bar on the worst gradient (5e-2) had been set around one sign
flip, and any change of a convolution's summation order moved which element flips.

    python tests/tools/mint_golden_nonsquare.py            (~10 min on 8 cores)
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "tools")]
sys.dont_write_bytecode = True
import helpers as H  # noqa: E402
import mint_golden as MG  # noqa: E402
import mint_golden_large as ML  # noqa: E402
from oracle import dmlnet_ref as O  # noqa: E402

SEED, SHAPE = 11, (2, 3, 128, 192)


def main():
    torch.set_num_threads(8)
    MG.install_shims()
    sys.path.insert(0, os.path.join(MG.REF, "DeepLabV3Plus-Pytorch"))
    import network as R  # the reference package

    def prep(m):
        m.train()
        m.classifier.aspp.project[3].eval()
        O.set_bn_momentum(m.backbone, 0.01)

    print("G13N non-square input %s, conditioned weights (seed %d)" % (SHAPE, SEED))
    ctor = lambda: R.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)  # noqa: E731
    shapes = H.shapes_of(ctor())
    img = H.synth_tensor(SEED, "g13n.img", SHAPE)
    sd, (bidx, bval), proof = ML.condition(ctor, shapes, SEED, img, prep)
    chk = H.conditioned_state_dict(shapes, SEED, bidx.numpy(), bval.numpy())
    assert chk.keys() == sd.keys() and all(torch.equal(chk[k], sd[k]) for k in sd)
    # the oracle on the same weights reproduces the reference (the oracle is what the test runs beside the HIP model)
    ref, orc = ctor(), O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    for m in (ref, orc):
        m.load_state_dict(sd)
        prep(m)
    with torch.no_grad():
        MG.assert_close(orc(img)[0], ref(img)[0], 1e-4, "logits: oracle vs reference")
    MG.save("g13n_nonsquare", seed=SEED, shape=torch.tensor(SHAPE), beta_idx=bidx, beta_val=bval, **proof)


if __name__ == "__main__":
    main()
