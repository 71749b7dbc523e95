import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.randn(8192, 8192, device="cuda")
sums = torch.zeros(5, dtype=torch.float64, device="cuda")
dist.all_reduce(sums); torch.cuda.synchronize()
def busy():
    y = x
    for _ in range(20): y = y @ x
    return y
for name, fn in (("sums[4] = 16.0", lambda: sums.__setitem__(4, 16.0)), ("all_reduce(sums)", lambda: dist.all_reduce(sums)),
                 ("all_reduce big", lambda: dist.all_reduce(x)), ("all_reduce async_op", lambda: dist.all_reduce(sums, async_op=True))):
    torch.cuda.synchronize(); busy(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-22s host %.3f ms (queue drained after %.1f ms)" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
