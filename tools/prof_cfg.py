import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip.configs import baseline_config
name = sys.argv[1]
flow, x = baseline_config(name)
for nrm in flow.getNormalizers():
    if hasattr(nrm, "nb_steps"): nrm.nb_steps = 20
for _ in range(3):
    for p in flow.parameters(): p.grad = None
    z, ld = flow(x); flow.loss(z, ld).backward()
torch.cuda.synchronize()
