"""Random Monotonic-normalizer shapes against the CPU oracle: forward (z, jac), every gradient, the bisection inverse (plain and
scattered).  One-off validation after changes to the weight pack or the kernel dispatch -- the committed tests pin the shapes
the reference uses; this walks the space between them.   python tests/fuzz_mono.py [n_cases] [seed]
(lives under tests/ because it uses oracle/ as the checker; tests/test_gpu_fuzz.py runs a short fixed-seed walk of it)"""
import os, sys, random, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd"), os.path.join(ROOT, "tests")]
from oracle import gnf_oracle as O          # noqa: E402
from models import MonotonicNormalizer       # noqa: E402
DEV = "cuda:0"
TOL, GTOL = 1e-5, 1e-4


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def one(case, rng, fixed=None):
    if fixed is not None:
        return run_case(case, *fixed)
    nl = rng.choice([1, 2, 2, 3, 3, 3, 4])
    kind = rng.random()
    if kind < .35:
        w = rng.randint(1, 208); hidden = [w] * nl                                  # equal widths (the reference's nets)
    elif kind < .7:
        base = rng.choice([48, 49, 50, 51, 52, 64, 65, 96, 97, 100, 111, 112, 113, 144, 145, 150, 159, 160, 161, 200])
        hidden = [base] * nl
    else:
        hidden = [rng.randint(1, 208) for _ in range(nl)]
    c = rng.choice([1, 2, 5, 16, 17, 30, 30, 30, 32, 33, 40])
    S = rng.choice([1, 2, 3, 7, 20, 20, 20, 21, 31, 32, 40])
    B, d = rng.randint(1, 40), rng.randint(1, 12)
    return run_case(case, hidden, c, S, B, d)


def run_case(case, hidden, c, S, B, d):
    torch.manual_seed(case)
    norm = MonotonicNormalizer(hidden, c, nb_steps=S, solver="CC")
    x = torch.randn(B, d) * 1.5
    h = torch.randn(B, d, c)
    ps = [p.detach().clone() for p in norm.integrand_net.flat_params()]
    layers = [(ps[i].requires_grad_(True), ps[i + 1].requires_grad_(True)) for i in range(0, len(ps), 2)]
    xr, hr = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
    z0, j0 = O.monotonic_forward(xr, hr, layers, S)
    gz, gj = torch.randn(B, d), torch.randn(B, d)
    ((z0 * gz).sum() + (torch.log(j0) * gj).sum()).backward()
    norm = norm.to(DEV)
    xg, hg = x.to(DEV).requires_grad_(True), h.to(DEV).requires_grad_(True)
    z, jac = norm(xg, hg)
    ((z * gz.to(DEV)).sum() + (torch.log(jac) * gj.to(DEV)).sum()).backward()
    errs = {"z": rel(z.cpu(), z0.detach()), "jac": rel(jac.cpu(), j0.detach()),
            "dx": rel(xg.grad.cpu(), xr.grad), "dh": rel(hg.grad.cpu(), hr.grad)}
    for k, ((W, b), pw, pb) in enumerate(zip(layers, norm.integrand_net.flat_params()[0::2], norm.integrand_net.flat_params()[1::2])):
        errs["dW%d" % k] = rel(pw.grad.cpu(), W.grad)
        errs["db%d" % k] = rel(pb.grad.cpu(), b.grad)
    # the output bias is ONE number, a sum of signed terms bounded by |gz| |x| / 2 * 2 + |gj| / 1.05 each (ELU' <= 1, f >= 1.05):
    # measured against that bound, not against the sum itself (six elements can cancel to 1e-3 of their terms)
    kl = "db%d" % (len(layers) - 1)
    bound = float((gz.abs() * x.abs() + gj.abs()).sum())
    errs[kl] = min(errs[kl], float((norm.integrand_net.flat_params()[-1].grad.cpu().double() - layers[-1][1].grad.double()).abs().max())
                   / max(bound, 1e-30) * 10.)
    bad = [k for k, v in errs.items() if not (v < (TOL if k in ("z", "jac") else GTOL))]
    if bad:
        # knife edges: a hidden ReLU pre-activation within a few fp32 ulps of zero (fp64 evaluation, tests/conftest.py) flips a
        # gate in one of the two fp32 computations and moves a gradient by that element's whole share.  Such elements get a zero
        # cotangent and both sides are evaluated again -- what still differs is a defect.
        from conftest import integrand_knife_elements
        knife = integrand_knife_elements(x, h, [(W.detach(), b_.detach()) for W, b_ in layers], S)
        nk = int(knife.sum())
        if nk:
            gz2, gj2 = gz.masked_fill(knife, 0.), gj.masked_fill(knife, 0.)
            for t in [xr, hr] + [p for pair in layers for p in pair]:
                t.grad = None
            z0, j0 = O.monotonic_forward(xr, hr, layers, S)
            ((z0 * gz2).sum() + (torch.log(j0) * gj2).sum()).backward()
            for t in [xg, hg] + list(norm.integrand_net.flat_params()):
                t.grad = None
            z, jac = norm(xg, hg)
            ((z * gz2.to(DEV)).sum() + (torch.log(jac) * gj2.to(DEV)).sum()).backward()
            errs2 = {"dx": rel(xg.grad.cpu(), xr.grad), "dh": rel(hg.grad.cpu(), hr.grad)}
            for k, ((W, b_), pw, pb) in enumerate(zip(layers, norm.integrand_net.flat_params()[0::2],
                                                      norm.integrand_net.flat_params()[1::2])):
                errs2["dW%d" % k] = rel(pw.grad.cpu(), W.grad)
                errs2["db%d" % k] = rel(pb.grad.cpu(), b_.grad)
            bad = ["%s %.1e (%d knife elements zeroed)" % (k, v, nk) for k, v in errs2.items() if not v < GTOL]
            bad += [k for k in ("z", "jac") if not errs[k] < TOL]
    # inverse: round trip through the kernel's own forward, and the scattered form == the plain one
    with torch.no_grad():
        zt = z.detach()
        xi = norm.inverse_transform(zt, hg.detach())
        if float((xi - xg.detach()).abs().max()) > 2e-3 and float(x.abs().max()) < 19.:
            bad.append("inverse %.2e" % float((xi - xg.detach()).abs().max()))
        if hasattr(norm, "inverse_transform_into"):
            out = torch.full((d, 64), 3.25, device=DEV)
            cols = torch.randperm(64)[:B].to(torch.int32).to(DEV)
            if norm.inverse_transform_into(zt, hg.detach(), out, cols):
                want = torch.full((d, 64), 3.25, device=DEV)
                want[:, cols.long()] = xi.t()
                if not torch.equal(out, want):
                    bad.append("scatter")
    return hidden, c, S, B, d, errs, bad


def walk(n, seed):
    """n random cases from `seed`: list of (case, description, failures)"""
    rng = random.Random(seed)
    out = []
    for case in range(n):
        hidden, c, S, B, d, errs, bad = one(case, rng)
        out.append((case, "hidden %s c %d S %d B %d d %d" % (hidden, c, S, B, d), bad))
    return out


def main():
    fixed = None
    if len(sys.argv) > 1 and sys.argv[1] == "--shape":      # --shape "65,65" c S B d [n seeds]: one shape, other data
        fixed = ([int(v) for v in sys.argv[2].split(",")],) + tuple(int(v) for v in sys.argv[3:7])
        n = int(sys.argv[7]) if len(sys.argv) > 7 else 8
        rng = random.Random(0)
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
        rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails = 0
    for case in range(n):
        try:
            hidden, c, S, B, d, errs, bad = one(1000 + case if fixed else case, rng, fixed)
        except Exception:                                                           # noqa: BLE001
            fails += 1
            print("case %d raised:\n%s" % (case, traceback.format_exc()), flush=True)
            continue
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("case %3d hidden %-22s c %2d S %2d B %2d d %2d  worst %s %.1e %s" % (case, hidden, c, S, B, d, worst[0], worst[1],
                                                                                 ("FAIL " + ",".join(bad)) if bad else "ok"), flush=True)
        fails += bool(bad)
    print("%d cases, %d failures" % (n, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
