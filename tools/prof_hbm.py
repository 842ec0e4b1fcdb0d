"""Micro-driver for rocprofv3 --pmc: the HBM-bound kernels (fused affine fwd/bwd at 1e6 x 63, DAG gate fwd at cfg4)."""
import sys
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import ops
dev = "cuda:0"
torch.manual_seed(0)
B, d = 1000000, 63
x = torch.randn(B, d, device=dev, requires_grad=True)
h = torch.randn(B, d, 2, device=dev, requires_grad=True)
for _ in range(3):
    z, _, ld, _ = ops.AffineFn.apply(x, h, False, False)
    torch.autograd.grad((z, ld), (x, h), (torch.ones_like(z), torch.ones_like(ld)))
xx = torch.randn(100, 784, device=dev)
A = torch.rand(784, 784, device=dev)
with torch.no_grad():
    for _ in range(3):
        ops.DagGateFn.apply(xx, A, ops.IMP_SOFT, ops.GATE_GUMBEL, 0., 1., False, None, None, 1, 1)
torch.cuda.synchronize()
