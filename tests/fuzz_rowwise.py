"""Random shapes / layouts through the row-wise kernels (gnf_rowwise.hip): the Affine normalizer (AffineNormalizer.py:9-17: forward
with log-det, backward, inverse; h contiguous [B, d, 2] or the MADE layout [B, 2 d] viewed as [B, d, 2]; values beyond both
clamps), the log-sum of a Jacobian's rows, the standard-normal log-density, and the one-launch loss -- against the oracle and
fp64 torch.   python tests/fuzz_rowwise.py [n] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd"), os.path.join(ROOT, "tests")]
from oracle import gnf_oracle as O      # noqa: E402
from gnf_hip import ops                  # noqa: E402
from models import AffineNormalizer      # noqa: E402
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def one(case, rng):
    g = torch.Generator().manual_seed(case)
    B = rng.choice([1, 2, 3, 7, 64, 100, 257, 1000, 4099])
    d = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 63, 64, 65, 127, 200, 784])
    if B * d > 1 << 20:
        B = max(1, (1 << 20) // d)
    made = rng.random() < .5
    x = torch.randn(B, d, generator=g) * 2.
    hraw = torch.randn(B, 2 * d, generator=g) * 3.                       # beyond the clamps (-5, 5) / (-5, 2) now and then
    def hview(t):
        return t.view(B, 2, d).permute(0, 2, 1) if made else t.view(B, d, 2)
    xr, hr = x.clone().requires_grad_(True), hraw.clone().requires_grad_(True)
    z0, j0 = O.affine_forward(xr, hview(hr))
    gz, gl = torch.randn(B, d, generator=g), torch.randn(B, generator=g)
    ld0 = torch.log(j0).sum(1)
    ((z0 * gz).sum() + (ld0 * gl).sum()).backward()
    nrm = AffineNormalizer()
    xg, hg = x.to(DEV).requires_grad_(True), hraw.to(DEV).requires_grad_(True)
    fused = getattr(nrm, "forward_logdet", None)
    if fused is not None:
        z, ld = fused(xg, hview(hg).clone() if False else hview(hg), None)
    else:
        z, jac = nrm(xg, hview(hg)); ld = ops.LogSumRowsFn.apply(jac)
    ((z * gz.to(DEV)).sum() + (ld * gl.to(DEV)).sum()).backward()
    errs = {"z": rel(z, z0), "logdet": rel(ld, ld0), "gx": rel(xg.grad, xr.grad), "gh": rel(hg.grad, hr.grad)}
    with torch.no_grad():
        xi = nrm.inverse_transform(z.detach(), hview(hg.detach()))
        errs["inverse"] = rel(xi, O.affine_inverse(z0.detach(), hview(hraw)))
        # log-density and the one-launch loss on this z / log-det
        zd, ldd = z.detach(), ld.detach()
        logn = ops.NormalLogDensityFn.apply(zd)
        errs["logN"] = rel(logn, O.normal_log_density(zd.cpu()))
        want = -float((ldd.double().cpu() + O.normal_log_density(zd.cpu()).double()).mean())
        if ops.nll_loss_fits(zd):
            got = float(ops.NllLossFn.apply(zd, ldd, None))
            errs["loss"] = abs(got - want) / max(1., abs(want))
    # jac rows through the generic reduction
    jr = (torch.rand(B, d, generator=g) + .05).requires_grad_(True)
    (torch.log(jr).sum(1) * gl).sum().backward()
    jg = jr.detach().to(DEV).requires_grad_(True)
    lsr = ops.LogSumRowsFn.apply(jg)
    (lsr * gl.to(DEV)).sum().backward()
    errs["logsum"] = rel(lsr, torch.log(jr).sum(1)); errs["g logsum"] = rel(jg.grad, jr.grad)
    # (the inverse subtracts two numbers that may nearly cancel, (z - mu) / sigma: 1e-4 of the tensor's max, as the gradients)
    bad = [k for k, v in errs.items() if not v < (1e-4 if k.startswith("g") or k == "inverse" else 1e-5)]
    return "B %5d d %3d %s" % (B, d, "made  " if made else "contig"), errs, bad


def walk(n, seed):
    rng = random.Random(seed)
    return [(case,) + one(case, rng) for case in range(n)]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    fails = 0
    for case, desc, errs, bad in walk(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0):
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("case %3d %s worst %s %.1e %s" % (case, desc, worst[0], worst[1], ("FAIL " + ",".join(bad)) if bad else "ok"), flush=True)
        fails += bool(bad)
    print("%d cases, %d failures" % (n, fails))
    sys.exit(1 if fails else 0)
