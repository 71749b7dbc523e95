"""Authoring container only: mints tests/golden/g13_cityscapes_labels.npz from the reference's own Cityscapes class
(datasets/cityscapes.py).  Needs /root/reference; the GPU box never runs this."""
import contextlib, importlib.util, io, os, sys, types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/DeepLabV3Plus-Pytorch/datasets/cityscapes.py"
sys.dont_write_bytecode = True
# the module imports torchvision.transforms and matplotlib at the top; neither is used by encode_target
for name in ("torchvision", "torchvision.transforms", "matplotlib", "matplotlib.pyplot"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
if not hasattr(sys.modules["torchvision"], "transforms"):
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
spec = importlib.util.spec_from_file_location("ref_cityscapes", REF)
mod = importlib.util.module_from_spec(spec)
with contextlib.redirect_stdout(io.StringIO()):
    spec.loader.exec_module(mod)
C = mod.Cityscapes

rng = np.random.default_rng(13)
raw = rng.integers(0, 34, size=(3, 37, 53)).astype(np.uint8)
raw[0, 0, :34] = np.arange(34, dtype=np.uint8)         # every raw id at least once
out = {"raw": raw}
for tag, unk in (("none", None), ("shipped", [14, 15]), ("train16", [13, 14, 15]), ("one", [18]), ("first", [0, 5])):
    C.unknown_target = unk
    t, tt = C.encode_target(raw)
    out["target_" + tag] = np.asarray(t, dtype=np.int64)
    out["true_" + tag] = np.asarray(tt, dtype=np.int64)
    out["unk_" + tag] = np.asarray(unk if unk is not None else [], dtype=np.int64)
# evaluation-time relabel, test_embedding.py:448-451, executed literally on the 'shipped' encoding
lab = out["target_shipped"].copy()
lab[lab == 13] = -1
lab[lab >= 14] -= 1
lab[lab == -1] = 16
lab[lab == 254] = 255
out["eval_relabel_shipped"] = lab
out["train_id_to_color"] = np.asarray(C.train_id_to_color)
out["id_to_train_id"] = np.asarray(C.id_to_train_id)
dec_in = out["target_shipped"][0].copy()
out["decoded_shipped0"] = np.asarray(C.decode_target(dec_in.copy()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_cityscapes_labels.npz"), **out)
print({k: v.shape for k, v in out.items()})
