"""GPU box, under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`: NL launches of ONE conv shape on rotating operand sets
(so that nothing is cache-resident between launches), for the HBM bytes a single launch really moves.
argv: H C N k mode(0 fwd | 1 dgrad) [accum]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
H, Cc, N, k, mode = (int(v) for v in sys.argv[1:6])
accum = int(sys.argv[6]) if len(sys.argv) > 6 else 0
B, NL = 16, 12
M = B * H * H
w = (torch.randn(N, k, k, Cc, device="cuda") * 0.05).to(torch.bfloat16)
sets = [(torch.randn(B, H, H, Cc, device="cuda").to(torch.bfloat16), torch.zeros(B, H, H, N, device="cuda", dtype=torch.bfloat16))
        for _ in range(NL)]
torch.cuda.synchronize()
for x, y in sets:
    d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B,
                 Hi=H, Wi=H, C=Cc, ldx=Cc, Ho=H, Wo=H, N=N, ldy=N, R=k, S=k, stride=1, dil=1, pad=k // 2, dtype=1, y_f32=0,
                 accum=accum, mode=mode)
    assert lib.dml_conv_igemm(C.byref(d), st) == 0
torch.cuda.synchronize()
print("algorithmic per launch: read %.1f MB (x %.1f + w %.2f%s), write %.1f MB" % (
    (M * Cc * 2 + N * k * k * Cc * 2 + (M * N * 2 if accum else 0)) / 1e6, M * Cc * 2 / 1e6, N * k * k * Cc * 2 / 1e6,
    " + y %.1f" % (M * N * 2 / 1e6) if accum else "", M * N * 2 / 1e6))
