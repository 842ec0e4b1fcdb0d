"""Micro-driver for rocprofv3: the fused MNISTCNN conv front (fwd+bwd) and the Monotonic
normalizer (fwd+bwd) at the cfg4 size (78 400 masked images / elements)."""
import sys
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import ops
which = sys.argv[1] if len(sys.argv) > 1 else "cnn"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
dev = "cuda:0"
if which == "cnn":
    # round 5: the conv front AS THE HEADLINE STEP RUNS IT -- gate + conv front as one autograd node (ops.DagConvFrontFn), x
    # frozen, A = MNIST_A_prior(28, 2) trainable: Philox Gumbel gate -> cnn_fwd_wino_k -> cnn_bwd_wino_k with the compact de
    # of the column plan -> the (i, slot)-threaded gate backward.  ("cnn_dense": the separate nodes with a dense de)
    from models.NormalizingFlowFactories import MNIST_A_prior
    B = 100
    x = torch.randn(B, 784, device=dev)
    A = MNIST_A_prior(28, 2).to(dev).requires_grad_(True)
    W1, b1 = (torch.randn(16, 1, 3, 3, device=dev) * .3).requires_grad_(True), (torch.randn(16, device=dev) * .1).requires_grad_(True)
    W2, b2 = (torch.randn(16, 16, 3, 3, device=dev) * .1).requires_grad_(True), (torch.randn(16, device=dev) * .1).requires_grad_(True)
    gp = torch.randn(B * 784, 2304, device=dev)
    for it in range(iters):
        out = ops.dag_conv_front(x, A, ops.IMP_SOFT, ops.GATE_GUMBEL, 0., 1., None, None, 1234, it + 1, W1, b1, W2, b2)
        out.backward(gp)
elif which == "cnn_dense":
    n = 78400
    e = (torch.randn(n, 784, device=dev) * (torch.rand(n, 784, device=dev) < .03).float()).requires_grad_(True)
    W1, b1 = (torch.randn(16, 1, 3, 3, device=dev) * .3).requires_grad_(True), (torch.randn(16, device=dev) * .1).requires_grad_(True)
    W2, b2 = (torch.randn(16, 16, 3, 3, device=dev) * .1).requires_grad_(True), (torch.randn(16, device=dev) * .1).requires_grad_(True)
    gp = torch.randn(n, 2304, device=dev)
    for _ in range(iters):
        out = ops.MnistConvFn.apply(e, W1, b1, W2, b2)
        out.backward(gp)
else:
    from models import MonotonicNormalizer
    B, d, c = 100, 784, 30
    norm = MonotonicNormalizer([50, 50, 50], c, nb_steps=20).to(dev)
    x = torch.randn(B, d, device=dev).requires_grad_(True)
    h = torch.randn(B, d, c, device=dev).requires_grad_(True)
    for _ in range(iters):
        z, jac = norm(x, h)
        (z.sum() + jac.sum()).backward()
torch.cuda.synchronize()
print("done")
