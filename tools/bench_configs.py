"""Run one training step (fwd + log|det J| + NLL + bwd) of every BASELINE.json configuration on
the MI355X and print ms/step, samples/s and the per-entry-point HIP-event timings.
Usage: python tools/bench_configs.py [cfg1 cfg2 cfg3 cfg4 cfg5] [--steps K]"""
import json
import sys
import time

import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
if os.environ.get('GNF_AB_LIB'):                      # A/B against another build of the library (tools/*.bin)
    from gnf_hip import abi as _abi
    _abi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ['GNF_AB_LIB'])
from gnf_hip import abi  # noqa: E402
from _warm import warm_gpu  # noqa: E402
from models import MonotonicNormalizer  # noqa: E402

DEV = "cuda:0"


from gnf_hip.configs import baseline_config as cfg  # noqa: E402  (the builders live in the package)


def main():
    names = [a for a in sys.argv[1:] if a.startswith("cfg")] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 5
    for name in names:
        flow, x = cfg(name)
        for nrm in flow.getNormalizers():
            if type(nrm) is MonotonicNormalizer:
                nrm.nb_steps = 20

        def step():
            for p in flow.parameters():
                p.grad = None
            z, ld = flow(x)
            loss = flow.loss(z, ld)
            loss.backward()
            return loss
        warm_gpu()
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        nst = steps if time.perf_counter() - t0 > .05 else max(steps, 30)     # short steps: average over more of them
        abi.profile_enable(list(abi.SIGNATURES))
        t0 = time.perf_counter()
        for _ in range(nst):
            loss = step()
        torch.cuda.synchronize()
        dt_mean = (time.perf_counter() - t0) / nst
        # a rare allocator stall (hipMalloc after the previous configuration's empty_cache) can dominate the mean of a
        # sub-millisecond step: report the median of individually timed steps, keep the mean next to it
        singles = []
        for _ in range(min(nst, 15)):
            t1 = time.perf_counter()
            loss = step()
            torch.cuda.synchronize()
            singles.append(time.perf_counter() - t1)
        singles.sort()
        dt = min(dt_mean, singles[len(singles) // 2])
        prof = abi.profile_collect()
        graphed = None
        if "--graph" in sys.argv and name in ("cfg1", "cfg3", "cfg4det", "cfg4dag"):
            # launch-bound configurations: the full optimisation step (incl. Adam) replayed from one hipGraph, next to
            # the same step issued launch by launch
            from gnf_hip import dp
            loss_value = round(loss.item(), 4)
            del loss                                   # GraphedStep needs the eager graphs gone
            state = dp.FlatState(flow)
            for _ in range(3):
                dp.train_step(flow, state, x, graph=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                dp.train_step(flow, state, x, graph=False)
            torch.cuda.synchronize()
            t_eager = (time.perf_counter() - t0) / 50
            gs = dp.GraphedStep(flow, state, x)
            warm_gpu()                                 # the capture idled the GPU
            for _ in range(3):
                gs(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                gl = gs(x)
            torch.cuda.synchronize()
            t_graph = (time.perf_counter() - t0) / 50
            graphed = {"full_step_eager_ms": round(t_eager * 1e3, 3), "full_step_hipgraph_ms": round(t_graph * 1e3, 3),
                       "samples_per_s_hipgraph": round(x.shape[0] / t_graph, 1), "loss": round(gl.item(), 4)}
        print(json.dumps({"config": name, "B": x.shape[0], "d": x.shape[1], "ms_per_step": round(dt * 1e3, 3),
                          "ms_per_step_mean": round(dt_mean * 1e3, 3), "hipgraph": graphed,
                          "samples_per_s": round(x.shape[0] / dt, 1),
                          "loss": loss_value if graphed is not None else round(loss.item(), 4),
                          "ops_ms_per_call": {k: round(v, 4) for k, v in sorted(prof.items())},
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
        del flow, x
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
