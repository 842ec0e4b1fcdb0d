"""Experiment (round 6): the DAG acyclicity term of cfg4 (six d x d library GEMMs + 4 small launches, ~0.12 ms of a 5.74 ms step; it depends on
A only) launched on a SIDE stream at the top of the step instead of in flow.loss() behind the forward.  Measures whether the hardware finds
room for it next to the forward's kernels.    python tools/exp_side_stream_loss.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import dp, ops
from gnf_hip.configs import baseline_config
from models.Conditionners.DAGConditioner import DAGConditioner
from _warm import warm_gpu

warm_gpu()
flow, x = baseline_config("cfg4")
state = dp.FlatState(flow)
cond = flow.getConditioners()[0]
orig_loss = DAGConditioner.loss
side = torch.cuda.Stream()
box = {}


def step(mode):
    if mode == "side":
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            box["val"] = orig_loss(cond)
            box["ev"] = torch.cuda.Event(); box["ev"].record(side)

        def patched(self):
            torch.cuda.current_stream().wait_event(box["ev"])
            box["val"].record_stream(torch.cuda.current_stream())
            return box["val"]
        DAGConditioner.loss = patched
    else:
        DAGConditioner.loss = orig_loss
    out = dp.train_step(flow, state, x, graph=False)
    DAGConditioner.loss = orig_loss
    return out


for mode in ("main", "side", "main", "side", "main", "side"):
    for _ in range(5):
        l = step(mode)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(40):
        l = step(mode)
    torch.cuda.synchronize()
    print("%s stream: %.3f ms per step   loss %.6f" % (mode, (time.perf_counter() - t) / 40 * 1e3, float(l)))
