"""Random sizes / modes through the DAG gate (gnf_dag_gate.hip behind gnf_hip.ops.DagGateFn and the column-plan pair of the fused
front): every importance form x gate form of DAGConditioner.py:94-166 with INJECTED noise, dimensions that are no multiple of
4, one-hot columns, sparse and dense A, against the oracle's masked inputs and their autograd.   python tests/fuzz_gate.py [n] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd"), os.path.join(ROOT, "tests")]
from oracle import gnf_oracle as O      # noqa: E402
from gnf_hip import ops                  # noqa: E402
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def one(case, rng):
    g = torch.Generator().manual_seed(case)
    d = rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 12, 16, 17, 31, 33, 63, 64, 70])
    B = rng.choice([1, 2, 3, 7, 16, 33])
    dens = rng.choice([1., 1., .5, .1])
    A = (torch.rand(d, d, generator=g) * 1.6 - .3) * (torch.rand(d, d, generator=g) < dens).float()
    x = torch.randn(B, d, generator=g)
    imp = rng.choice(["raw", "soft", "hard_soft", "hard_sq"])
    gate = rng.choice(["det", "gumbel", "gumbel", "noise"]) if imp != "raw" else "det"
    T = rng.choice([1., 1., .5, .7]) if gate == "gumbel" else 1.
    if imp == "hard_sq" and gate == "gumbel":          # importance = A^2 must stay a probability (the reference takes log(1 - A^2))
        A = A.clamp(-.95, .95)
    hot = rng.random() < .25
    h_thresh = rng.choice([.05, .3]) if imp.startswith("hard") else 0.
    s_thresh = imp in ("soft", "hard_soft")
    u1 = u2 = nz = None
    if gate == "gumbel":
        u1 = torch.rand(B, d, d, generator=g).clamp(1e-6, 1 - 1e-6)
        u2 = torch.rand(B, d, d, generator=g).clamp(1e-6, 1 - 1e-6)
    if gate == "noise":
        nz = torch.randn(B, d, d, generator=g)
    xr, Ar = x.clone().requires_grad_(True), A.clone().requires_grad_(True)
    if imp == "raw":
        e0 = O.dag_masked_inputs(xr, Ar, False, 0., False, False, 1., None, None, None, hot)
    else:
        e0 = O.dag_masked_inputs(xr, Ar, s_thresh, h_thresh, gate == "gumbel", gate == "noise", T, u1, u2, nz, hot)
    w = torch.randn(e0.shape, generator=g)
    (e0 * w).sum().backward()
    xg, Ag = x.to(DEV).requires_grad_(True), A.to(DEV).requires_grad_(True)
    im = {"raw": ops.IMP_RAW, "soft": ops.IMP_SOFT, "hard_soft": ops.IMP_HARD_SOFT, "hard_sq": ops.IMP_HARD_SQ}[imp]
    gm = {"det": ops.GATE_DET, "gumbel": ops.GATE_GUMBEL, "noise": ops.GATE_NOISE}[gate]
    a1 = (u1 if gate == "gumbel" else nz)
    e = ops.DagGateFn.apply(xg, Ag, im, gm, h_thresh, T, hot, a1.to(DEV) if a1 is not None else None,
                            u2.to(DEV) if u2 is not None else None, 0, 0)
    (e * w.to(DEV)).sum().backward()
    errs = {"e": rel(e, e0), "gx": rel(xg.grad, xr.grad), "gA": rel(Ag.grad, Ar.grad) if float(Ar.grad.abs().max()) > 0 else
            float(Ag.grad.abs().max())}
    # hard thresholds are step functions of A: an entry within rounding of the threshold may fall on either side
    near = False
    if imp.startswith("hard"):
        impv = O.dag_soft_thresholded_A(A) if s_thresh else A ** 2
        near = bool(((impv - h_thresh).abs() < 1e-6).any())
    bad = [] if near else [k for k, v in errs.items() if not v < (1e-5 if k == "e" else 1e-4)]
    if bad:
        # fp64 arbitration: the reference's own expressions lose digits in fp32 -- 2 (sigmoid(2 A^2) - .5) for a tiny A is a
        # difference of two numbers at .5 (4 % off at A = 1e-3), and a uniform within 1e-3 of 1 turns that into a visible
        # gate -- so two correct fp32 evaluations differ there; a defect shows as the kernel being much further from the
        # fp64 value than the fp32 oracle is
        x6, A6 = x.double().requires_grad_(True), A.double().requires_grad_(True)
        d6 = lambda t: t.double() if t is not None else None
        if imp == "raw":
            e6 = O.dag_masked_inputs(x6, A6, False, 0., False, False, 1., None, None, None, hot)
        else:
            e6 = O.dag_masked_inputs(x6, A6, s_thresh, h_thresh, gate == "gumbel", gate == "noise", T, d6(u1), d6(u2), d6(nz), hot)
        (e6 * w.double()).sum().backward()
        trio = {"e": (e, e0, e6), "gx": (xg.grad, xr.grad, x6.grad), "gA": (Ag.grad, Ar.grad, A6.grad)}
        bad = [k for k in bad if rel(trio[k][0], trio[k][2]) > 4. * rel(trio[k][1], trio[k][2]) + 1e-6]
    desc = "d %2d B %2d imp %-9s gate %-6s T %.1f hot %d dens %.1f" % (d, B, imp, gate, T, hot, dens)
    return desc, errs, bad


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
