"""Run one training step (fwd + log|det J| + NLL + bwd) of every BASELINE.json configuration on
the MI355X and print ms/step, samples/s and the per-entry-point HIP-event timings.
Usage: python tools/bench_configs.py [cfg1 cfg2 cfg3 cfg4 cfg5] [--steps K]"""
import json
import sys
import time

import torch

sys.path[:0] = ['/root/repo', '/root/repo/graphical-normalizing-flows_amd']
from gnf_hip import abi  # noqa: E402
from models import (buildFCNormalizingFlow, CouplingConditioner, AutoregressiveConditioner, DAGConditioner,  # noqa: E402
                    AffineNormalizer, MonotonicNormalizer)
from models.NormalizingFlowFactories import buildMNISTNormalizingFlow  # noqa: E402
import bench  # noqa: E402

DEV = "cuda:0"


def cfg(name):
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(0)
    if name == "cfg1":      # toy 8gaussians (lib/toy_data.py:81-98 restated), Affine+Coupling
        B = 512
        ang = torch.randint(0, 8, (B,), generator=g).float() * (3.141592653589793 / 4)
        x = (torch.stack((torch.cos(ang), torch.sin(ang)), 1) * 4 + torch.randn(B, 2, generator=g) * .5) / 1.414
        f = buildFCNormalizingFlow(1, CouplingConditioner, {"in_size": 2, "hidden": [150, 150], "out_size": 150},
                                   AffineNormalizer, {})
    elif name == "cfg2":    # POWER d=6 Monotonic+DAG (UCIExperimentsConfigurations.yml:1-14, UCI:83-93)
        x = torch.randn(10000, 6, generator=g)
        f = buildFCNormalizingFlow(1, DAGConditioner, {"in_size": 6, "hidden": [60, 60, 60], "out_size": 30, "l1": 0.,
                                                       "gumble_T": .5, "nb_epoch_update": 30, "hot_encoding": True},
                                   MonotonicNormalizer, {"integrand_net": [100, 100, 100], "cond_size": 30,
                                                         "nb_steps": 20, "solver": "CC"})
    elif name == "cfg3":    # MNIST d=784 Affine+Autoregressive 1024^3
        x = bench.pseudo_mnist(g, 100, 784)
        f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 784, "hidden": [1024] * 3, "out_size": 2},
                                   AffineNormalizer, {})
    elif name == "cfg4":
        x = bench.pseudo_mnist(g, 100, 784)
        f = bench.build_flow()
    elif name == "cfg4det":  # cfg4 after the DAG phase: post_process() froze a binary A, the gate is deterministic
        x = bench.pseudo_mnist(g, 100, 784)
        f = bench.build_flow()
        for c in f.getConditioners():
            with torch.no_grad():
                c.post_process(zero_threshold=.1)
    elif name == "cfg4dag":  # the state update_dual_param() ends in: an acyclic binary A (window parents that precede the
        x = bench.pseudo_mnist(g, 100, 784)          # pixel in raster order), post-processed, dag_const = l1 = 0
        f = bench.build_flow()
        for c in f.getConditioners():
            with torch.no_grad():
                idx = torch.arange(784)
                c.A.mul_((idx[None, :] < idx[:, None]).float())
                c.post_process(zero_threshold=.1)
                c.dag_const = torch.tensor(0.)
                c.l1_weight = torch.tensor(0.)
                c.is_invertible = True
    elif name == "cfg5":    # BSDS300 d=63 synthetic (yml:347-358), B=50000
        x = torch.randn(50000, 63, generator=g)
        f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 63, "hidden": [630] * 3, "out_size": 30},
                                   MonotonicNormalizer, {"integrand_net": [150, 150, 150], "cond_size": 30,
                                                         "nb_steps": 20, "solver": "CCParallel"})
    else:
        raise KeyError(name)
    return f.to(DEV), x.to(DEV)


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
