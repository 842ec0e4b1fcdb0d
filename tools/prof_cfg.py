import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/graphical-normalizing-flows_amd', '/root/repo/tools']
import bench_configs as bc
name = sys.argv[1]
flow, x = bc.cfg(name)
for nrm in flow.getNormalizers():
    if hasattr(nrm, "nb_steps"): nrm.nb_steps = 20
for _ in range(3):
    for p in flow.parameters(): p.grad = None
    z, ld = flow(x); flow.loss(z, ld).backward()
torch.cuda.synchronize()
