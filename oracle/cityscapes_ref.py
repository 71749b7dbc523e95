"""CPU restatement of the reference's Cityscapes label encoding -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(datasets/cityscapes.py -> dml_label_encode / dml_aug_apply_encoded in libdmlnet_hip.so) never does.

Follows datasets/cityscapes.py:132-154 of the reference (Cityscapes.encode_target): raw label id -> train id through the
class table (:27-63; ids 0..33, the 35th entry has id -1 and is reached by nothing a uint8 PNG can hold), then for each
entry h of `unknown_target` in order (cont = how many were removed before it): pixels equal to h - cont are parked at 100
and every id above moves down by one; parked pixels end as 255.  `target_true` is the plain train-id map.
Also the evaluation-time relabel of test_embedding.py:448-451 (the held-out class, id 13 after the shift, becomes 16 and
ids 14..16 move down, 254 -> 255).

Parity pin: tests/golden/g13_cityscapes_labels.npz, minted from the reference's own class by
tests/tools/mint_golden_labels.py (unknown_target None, [14, 15] as shipped, [13, 14, 15] as the README asks for training).
"""
import numpy as np

# (raw id, train id) of the public cityscapesScripts table, ids 0..33
RAW_TO_TRAIN = [255, 255, 255, 255, 255, 255, 255, 0, 1, 255, 255, 2, 3, 4, 255, 255, 255, 5, 255, 6, 7, 8, 9, 10, 11, 12,
                13, 14, 15, 255, 255, 16, 17, 18]


def encode_target(raw, unknown_target):
    table = np.asarray(RAW_TO_TRAIN + [255], dtype=np.int64)          # + the id -1 entry ('license plate')
    target = table[np.asarray(raw)]
    target_true = target.copy()
    if unknown_target is not None:
        cont = 0
        for h in unknown_target:
            target[target == h - cont] = 100
            for c in range(h - cont + 1, 19):
                target[target == c] = c - 1
            cont += 1
        target[target == 100] = 255
    return target, target_true


def eval_relabel(labels):
    """test_embedding.py:448-451, on an int64 array (in place semantics restated on a copy)."""
    lab = np.array(labels, dtype=np.int64)
    lab[lab == 13] = -1
    lab[lab >= 14] -= 1
    lab[lab == -1] = 16
    lab[lab == 254] = 255
    return lab
