import sys, torch
sys.path[:0]=["/root/repo/open-world-semantic-segmentation_amd"]
from dmlnet import _lib
lib=_lib.load(); st=torch.cuda.current_stream().cuda_stream
def timeit(fn,n=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (B,HW,C,ld) in ((16,2304,2048,2048),(16,2304,256,1280)):
    x=torch.randn(B,HW,ld,device="cuda").bfloat16(); o=torch.empty(B,C,device="cuda",dtype=torch.bfloat16)
    t=timeit(lambda: lib.dml_global_avgpool_fwd(x.data_ptr(),o.data_ptr(),B,HW,C,ld,1,st))
    print(B,HW,C,ld,"%.1f us"%t, "%.2f TB/s"%(B*HW*C*2/t/1e6))
