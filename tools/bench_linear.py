"""Small-batch Linear kernels (gnf_linear.hip) at the MADE layer shape of BASELINE cfg3 (100 x 1024 x 1024, GNF_LIN_SHAPE=M,N,K
for others): HIP-event time of each entry point with the mask as a degree rule / a tensor / absent, against the bytes a
layer has to stream (weights once per product: 4 N K).
python tools/bench_linear.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops
from _warm import warm_gpu  # noqa: E402
if os.environ.get('GNF_AB_LIB'):
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
dev = 'cuda:0'
M, N, K = [int(v) for v in os.environ.get('GNF_LIN_SHAPE', '100,1024,1024').split(',')]
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, requires_grad=True)
W = (torch.randn(N, K, device=dev) / K ** .5).requires_grad_(True)
b = torch.zeros(N, device=dev, requires_grad=True)
W2 = (torch.randn(N, N, device=dev) / N ** .5).requires_grad_(True)
b2 = torch.zeros(N, device=dev, requires_grad=True)
# MADE's natural-ordering degrees (AutoregressiveConditioner.py:85-87): hidden unit k has degree 783 - (k mod 784)
do, di = (783 - torch.arange(N, device=dev) % 784).float(), (783 - torch.arange(K, device=dev) % 784).float()
mask = (di[None, :] <= do[:, None]).float()
mask2 = (do[None, :] <= do[:, None]).float()
names = ("gnf_linear_fwd", "gnf_linear_bwd")           # bwd: both gradients of a layer, one launch
for kind in ("deg", "full", "none"):
    masks = None if kind == "none" else [mask, mask2]
    degs = [(do, di, False), (do, do, False)] if kind == "deg" else None
    def step():
        for t in (x, W, b, W2, b2): t.grad = None
        y = ops.mlp(x, [(W, b), (W2, b2)], masks, degs=degs)
        y.sum().backward()
    warm_gpu()
    for _ in range(5): step()
    abi.profile_enable(names)
    for _ in range(20): step()
    prof = abi.profile_collect()
    mb = 4 * N * K / 1e6
    print(kind, " ".join("%s %.1f us (%.2f TB/s of weights)" % (n[11:], prof[n] * 1e3, mb / prof[n] / 1e3) for n in names))
