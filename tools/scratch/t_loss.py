import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/graphical-normalizing-flows_amd']
import bench
DEV = "cuda:0"
flow = bench.build_flow().to(DEV)
cond = flow.getConditioners()[0]
def f():
    cond.A.grad = None
    cond.loss().backward()
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("constraint loss fwd+bwd ms (GPU timeline)", e0.elapsed_time(e1) / 50)
