import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
import torch.nn.functional as F
from gnf_hip import ops
torch.manual_seed(700)
n = 700
e = torch.randn(n, 784)
W1, b1 = torch.randn(16, 1, 3, 3) * .3, torch.randn(16) * .1
W2, b2 = torch.randn(16, 16, 3, 3) * .1, torch.randn(16) * .1
ps = [t.clone().requires_grad_(True) for t in (e, W1, b1, W2, b2)]
ref = torch.flatten(F.max_pool2d(F.conv2d(torch.relu(F.conv2d(ps[0].view(-1, 1, 28, 28), ps[1], ps[2])), ps[3], ps[4]), 2), 1)
gp = torch.randn(n, 2304)
(ref * gp).sum().backward()
pg = [t.clone().cuda().requires_grad_(True) for t in (e, W1, b1, W2, b2)]
out = ops.MnistConvFn.apply(*pg)
(out * gp.cuda()).sum().backward()
ge = pg[0].grad.cpu(); gr = ps[0].grad
err = (ge - gr).abs().amax(1) / gr.abs().amax(1)
bad = (err > 1e-4).nonzero().flatten()
print("bad images:", bad.tolist()[:40], "count", len(bad))
if len(bad):
    i = bad[0].item()
    d = (ge[i] - gr[i]).abs().view(28, 28)
    print("img", i, "bad pixels:", (d > 1e-4 * gr[i].abs().max()).nonzero().tolist()[:30])
for a, b, name in zip(pg[1:], ps[1:], ("W1", "b1", "W2", "b2")):
    print(name, ((a.grad.cpu() - b.grad).abs().max() / b.grad.abs().max()).item())
