"""Latency of the split bisection-inverse kernel on a level-sized problem (700 elements, H = 50, c = 30): time per call
for several node counts S.   python tools/bench_inv_small.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import ops, abi
if os.environ.get('GNF_AB_LIB'):
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
from models import MonotonicNormalizer
dev = 'cuda:0'
torch.manual_seed(0)
for H in (50, 64):
    norm = MonotonicNormalizer([H, H, H], 30, nb_steps=20).to(dev)
    params = [p.detach() for p in norm.integrand_net.parameters()]
    for n, d in ((100, 7), (100, 1), (1600, 7)):
        z = torch.randn(n, d, device=dev) * .3
        h = torch.randn(n, d, 30, device=dev)
        for S in (7, 20, 40):
            for _ in range(3):
                x = ops.monotonic_inverse(z, h, S, params)
            abi.profile_enable(("gnf_monotonic_inv",))
            for _ in range(20):
                x = ops.monotonic_inverse(z, h, S, params)
            t = abi.profile_collect()["gnf_monotonic_inv"]
            print("H=%d n=%5d S=%2d  inverse %.1f us  (%.2f us per bisection step)" % (H, n * d, S, t * 1e3, t * 1e3 / 20))
