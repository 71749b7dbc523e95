"""GPU box: the small streaming kernels of a train step (max pool, bilinear, input packing, fill) at the benchmark's
shapes under DML_GRID_CAP (child processes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path[:0] = [%r, %r]
import torch
from dmlnet import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream; bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 16
z0 = torch.randn(B, 384, 384, 64, device="cuda").to(bf); p0 = torch.empty(B, 192, 192, 64, device="cuda", dtype=bf)
am = torch.empty(B * 192 * 192 * 64, dtype=torch.uint8, device="cuda")
out = []
out.append(("maxpool_fwd", t(lambda: lib.dml_maxpool3x3s2_fwd(z0.data_ptr(), p0.data_ptr(), am.data_ptr(), B, 384, 384, 64, 1, st)), (z0.numel() + p0.numel()) * 2 + am.numel()))
dz0 = torch.empty_like(z0)
out.append(("maxpool_bwd", t(lambda: lib.dml_maxpool3x3s2_bwd(p0.data_ptr(), am.data_ptr(), dz0.data_ptr(), B, 384, 384, 64, 1, st)), (z0.numel() + p0.numel()) * 2 + am.numel()))
lo = torch.randn(B, 48, 48, 256, device="cuda").to(bf); hi = torch.empty(B, 192, 192, 256, device="cuda", dtype=bf)
out.append(("bilinear_fwd", t(lambda: lib.dml_bilinear_fwd(lo.data_ptr(), hi.data_ptr(), B, 48, 48, 192, 192, 256, 256, 256, 1, 0, 0, st)), (lo.numel() + hi.numel()) * 2))
out.append(("bilinear_bwd", t(lambda: lib.dml_bilinear_bwd(hi.data_ptr(), lo.data_ptr(), B, 48, 48, 192, 192, 256, 256, 256, 1, 0, 0, st)), (lo.numel() + hi.numel()) * 2))
img = torch.randn(B, 3, 768, 768, device="cuda"); x8 = torch.empty(B, 768, 768, 8, device="cuda", dtype=bf)
out.append(("pack_input", t(lambda: lib.dml_pack_input(img.data_ptr(), x8.data_ptr(), B, 3, 768, 768, 8, 1, st)), img.numel() * 4 + x8.numel() * 2))
g = torch.empty(58800000, device="cuda")
out.append(("fill 235 MB", t(lambda: lib.dml_fill_f32(g.data_ptr(), g.numel(), 0.0, st)), g.numel() * 4))
print(" | ".join("%%s %%.0f us %%.2f TB/s" %% (n, us, by / us / 1e6) for n, us, by in out))
''' % (ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"))
for cap in ("0", "8192", "32768", "131072"):
    env = dict(os.environ, DML_GRID_CAP=cap)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("CAP=%s: %s" % (cap, (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]), flush=True)
