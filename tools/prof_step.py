"""Micro-driver for rocprofv3: full optimisation steps (dp.train_step: fwd + loss + bwd + Adam) of a BASELINE
configuration, launch by launch (or, with --graph, as the replayed hipGraph).    python tools/prof_step.py cfg3 [steps] [--graph]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import dp
from gnf_hip.configs import baseline_config
flow, x = baseline_config(sys.argv[1])
for nrm in flow.getNormalizers():
    if hasattr(nrm, "nb_steps"): nrm.nb_steps = 20
state = dp.FlatState(flow)
graph = "--graph" in sys.argv
args = [a for a in sys.argv[1:] if a != "--graph"]
for _ in range(int(args[1]) if len(args) > 1 else 10):
    dp.train_step(flow, state, x, graph="auto" if graph else False)
torch.cuda.synchronize()
