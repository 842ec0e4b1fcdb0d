"""Monotonic normalizer kernels at the cfg4 element count (100 x 784, c = 30, S = 20; GNF_MONO_SHAPE=B,d,c,S for others):
forward / backward entry-point times for a list of hidden widths.    python tools/bench_mono.py 48 50 64"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
if os.environ.get('GNF_AB_LIB'):                      # A/B against another build of the library (tools/*.bin)
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
from models import MonotonicNormalizer
from _warm import warm_gpu  # noqa: E402
dev = 'cuda:0'
for H in [int(a) for a in sys.argv[1:]] or [50]:
    torch.manual_seed(0)
    B, d, c, S = [int(v) for v in os.environ.get('GNF_MONO_SHAPE', '100,784,30,20').split(',')]   # cfg5: 50000,63,30,20
    norm = MonotonicNormalizer([H, H, H], c, nb_steps=S).to(dev)
    x = torch.randn(B, d, device=dev, requires_grad=True); h = torch.randn(B, d, c, device=dev, requires_grad=True)
    def step():
        for p in norm.parameters(): p.grad = None
        x.grad = None; h.grad = None
        z, jac = norm(x, h)
        (z.sum() + torch.log(jac).sum()).backward()
        return z
    warm_gpu()
    for _ in range(3): step()
    abi.profile_enable(("gnf_monotonic_fwd", "gnf_monotonic_bwd"))
    for _ in range(10): z = step()
    prof = abi.profile_collect()
    print("H=%d  fwd %.4f ms  bwd %.4f ms  checksum %.6e" % (H, prof["gnf_monotonic_fwd"], prof["gnf_monotonic_bwd"], z.double().sum().item()))
