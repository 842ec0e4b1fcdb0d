"""per-parameter gradient errors of the Monotonic backward against the fp32 AND an fp64 CPU oracle (is a 1e-4 miss roundoff
of the fp32 oracle's own summation order, or a kernel defect?)   python tests/dbg_mono_wide_grads.py B d H,H,H"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/graphical-normalizing-flows_amd"]
from oracle import gnf_oracle as O
from models import MonotonicNormalizer

B, d = int(sys.argv[1]), int(sys.argv[2])
hidden = [int(v) for v in sys.argv[3].split(",")]
torch.manual_seed(B * 100 + d)
c, S = 30, 20
norm = MonotonicNormalizer(hidden, c, nb_steps=S)
x, h = torch.randn(B, d), torch.randn(B, d, c)
ps = [p.detach().cpu().clone() for p in norm.integrand_net.flat_params()]
gz, gj = torch.randn(B, d), torch.randn(B, d)


def oracle(dt):
    layers = [(ps[i].to(dt).clone().requires_grad_(True), ps[i + 1].to(dt).clone().requires_grad_(True)) for i in range(0, len(ps), 2)]
    xr, hr = x.to(dt).clone().requires_grad_(True), h.to(dt).clone().requires_grad_(True)
    z0, j0 = O.monotonic_forward(xr, hr, layers, S)
    ((z0 * gz.to(dt)).sum() + (j0 * gj.to(dt)).sum()).backward()
    return [xr.grad, hr.grad] + [t.grad for WB in layers for t in WB]


g32, g64 = oracle(torch.float32), oracle(torch.float64)
norm = norm.to("cuda:0")
xg, hg = x.cuda().detach().requires_grad_(True), h.cuda().detach().requires_grad_(True)
z, jac = norm(xg, hg)
((z * gz.cuda()).sum() + (jac * gj.cuda()).sum()).backward()
gg = [xg.grad, hg.grad] + [p.grad for p in norm.integrand_net.flat_params()]
names = ["x", "h"] + ["%s%d" % (k, i // 2) for i in range(len(ps)) for k in ("W" if i % 2 == 0 else "b",)]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


for n, a, b32, b64 in zip(names, gg, g32, g64):
    print("%-4s shape %-14s hip-vs-f64 %.2e   f32oracle-vs-f64 %.2e   hip-vs-f32oracle %.2e" % (n, tuple(a.shape), rel(a, b64), rel(b32, b64), rel(a, b32)))
