"""Training step (fwd + log-det + NLL + bwd) of the 3-scale MNIST factory flow (CNNormalizingFlow, scales 28/14/7,
kernel-2 priors, B = 100) with the Affine and the Monotonic normalizer.
Usage: python tools/bench_three_scale.py"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
import bench
from models import MonotonicNormalizer, AffineNormalizer
from models.NormalizingFlowFactories import buildMNISTNormalizingFlow
DEV = "cuda:0"
torch.manual_seed(0)
for norm, args in ((AffineNormalizer, {}), (MonotonicNormalizer, {"integrand_net": [50, 50, 50], "nb_steps": 20, "solver": "CC"})):
    f = buildMNISTNormalizingFlow([1, 1, 1], norm, args, l1=0., nb_epoch_update=10, hot_encoding=False, prior_kernel=2).to(DEV)
    g = torch.Generator().manual_seed(1)
    x = bench.pseudo_mnist(g, 100, 784).to(DEV)
    def step():
        for p in f.parameters(): p.grad = None
        z, ld = f(x); f.loss(z, ld).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print(norm.__name__, "3-scale step ms", (time.perf_counter() - t) / 5 * 1e3)
