"""How much bf16 storage (oracle/bf16_emu.py) and a 1e-3 relative perturbation of the image move the ORACLE's logits and
gradients: the noise floor the bf16 parity bars of tests/test_gpu_bf16_parity.py are set against.  CPU only.
    python tests/tools/bf16_noise.py 4x3x256x256 9
2 x 64 x 64: gradients 1 - cos 0.65 / 0.80 (noise); 4 x 256 x 256: 0.079 / 0.076."""
import sys, numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as H
from oracle import dmlnet_ref as O, bf16_emu
def run(emu, shape, seed, tag, perturb=0.0, dtype=torch.float32):
    o=O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed)); o=o.to(dtype); o.train(); o.classifier.aspp.project[3].eval()
    O.set_bn_momentum(o.backbone,0.01)
    if emu: bf16_emu.emulate_bf16_storage(o)
    img=H.synth_tensor(seed, tag+".img", shape).to(dtype)
    if perturb: img=img*(1+perturb*torch.randn(img.shape, generator=torch.Generator().manual_seed(3)).to(dtype))
    lab=H.synth_labels(seed, tag+".lab", (shape[0],shape[2],shape[3]),16,255,ignore_frac=0.05)
    lg,_,_=o(img); loss=O.dml_loss(lg,lab,0.01,255); loss.backward()
    return lg.detach().double(), {k:p.grad.double() for k,p in o.named_parameters()}
shape=tuple(int(x) for x in sys.argv[1].split('x')); seed=int(sys.argv[2])
a_lg,a=run(True,shape,seed,"t")
b_lg,b=run(False,shape,seed,"t")
c_lg,c=run(True,shape,seed,"t",perturb=1e-3)   # emu with a tiny input perturbation: chaos amplification at bf16 level
def stats(x,y):
    e=[]; cs=[]
    for k in x:
        sc=y[k].abs().max().item()+1e-30
        e.append((x[k]-y[k]).abs().max().item()/sc)
        u,v=x[k].flatten(),y[k].flatten(); cs.append(1-(u@v).item()/(u.norm().item()*v.norm().item()+1e-30))
    e=np.array(e); cs=np.array(cs)
    return "max-norm median %.2e p95 %.2e max %.2e | 1-cos median %.2e p95 %.2e max %.2e"%(np.median(e),np.percentile(e,95),e.max(),np.median(cs),np.percentile(cs,95),cs.max())
print("emu vs fp32: logits %.2e grads %s"%(H.rel_err(a_lg,b_lg), stats(a,b)))
print("emu vs emu(perturbed 1e-3 input): logits %.2e grads %s"%(H.rel_err(c_lg,a_lg), stats(c,a)))
