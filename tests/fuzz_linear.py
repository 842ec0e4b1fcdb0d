"""Random Linear + ReLU chains through gnf_hip.ops.mlp (the gnf_linear_* entry points: weight-streaming kernels for small
batches, the tall-layer kernels for M >= 2048 on narrow layers, the tiled GEMM otherwise; masks as tensors or as MADE degree
rules) against an fp64 torch autograd of F.linear(x, mask * W, b) -- AutoregressiveConditioner.py:24-25, DAGConditioner.py:7-20,
MLP.py:44-47.   python tests/fuzz_linear.py [n_cases] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd")]
from gnf_hip import ops       # noqa: E402
DEV = "cuda:0"
WIDTHS = [1, 2, 3, 12, 16, 30, 31, 32, 33, 50, 60, 63, 64, 65, 96, 100, 128, 129, 160, 255, 256, 630, 784, 1024, 1568, 2304]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def params(rng):
    r = rng.random()
    if r < .35:
        M = rng.choice([1, 2, 7, 16, 33, 64, 100, 127, 128])                 # small batch: the skinny kernels
    elif r < .55:
        M = rng.choice([129, 200, 512, 1000, 2047])
    else:
        M = rng.choice([2048, 2049, 4096, 10000, 60000])                      # tall batch
    nl = rng.choice([1, 2, 2, 3, 4])
    dims = [rng.choice(WIDTHS) if rng.random() < .7 else rng.randint(1, 300) for _ in range(nl + 1)]
    if M >= 2048:                                                             # keep the tall cases in memory and in time
        dims = [min(v, 256) for v in dims]
    mask_kind = rng.choice(["none", "none", "full", "deg", "deg_strict"])
    return M, nl, dims, mask_kind, rng.random() < .7


def one(case, rng, only=None):
    M, nl, dims, mask_kind, need_x = params(rng)
    if only is not None and case != only:
        return None
    g = torch.Generator().manual_seed(case)
    x = torch.randn(M, dims[0], generator=g)
    layers, masks, degs = [], [], []
    dprev = torch.randint(0, 9, (dims[0],), generator=g).float()
    for l in range(nl):
        K, N = dims[l], dims[l + 1]
        W, b = torch.randn(N, K, generator=g) / max(K, 1) ** .5, torch.randn(N, generator=g) * .1
        layers.append((W, b))
        if mask_kind != "none":
            strict = mask_kind == "deg_strict"
            do = torch.randint(0, 9, (N,), generator=g).float()
            m = (torch.lt if strict else torch.le)(dprev[None, :], do[:, None]).float()
            if mask_kind == "full":
                m = (torch.rand(N, K, generator=g) < .6).float()
            else:
                degs.append((do.to(DEV), dprev.to(DEV), strict))
            masks.append(m)
            dprev = do
    gy = torch.randn(M, dims[-1], generator=g)
    # fp64 reference.  Rows with a hidden pre-activation within the fp32 chain's error bound of zero get a zero cotangent: there an
    # fp32 chain and the fp64 one may gate differently, and ONE flipped gate moves that row's gradient by several per cent (at
    # 60 000 rows x 256 units a walk meets a handful of such rows per case).  The bound of a layer = 16 fp32 ulps of its own terms'
    # magnitude + the bound INHERITED from its inputs through |W|: behind a narrow bottleneck (walk 703, case 113: 255 -> 1 -> 198)
    # the single input carries the roundoff of a 255-term sum, which is far more than 16 ulps of its own small value
    xr = x.double().requires_grad_(need_x)
    ps = [(W.double().requires_grad_(True), b.double().requires_grad_(True)) for W, b in layers]
    a = xr
    knife = torch.zeros(M, dtype=torch.bool)
    aerr = torch.zeros(M, dims[0], dtype=torch.float64)          # bound on the fp32 chain's absolute error of the layer inputs
    for l, (W, b) in enumerate(ps):
        Wm = W * masks[l].double() if masks else W
        pre = torch.nn.functional.linear(a, Wm, b)
        if l < nl - 1:
            with torch.no_grad():
                mag = a.detach().abs() @ Wm.detach().abs().t() + b.detach().abs()
                aerr = 16 * 2. ** -23 * mag + aerr @ Wm.detach().abs().t()
                knife |= ((pre.detach().abs() < aerr) & (pre.detach() != 0)).any(1)
            a = torch.relu(pre)
        else:
            a = pre
    gy = gy * (~knife).float()[:, None]
    (a * gy.double()).sum().backward()
    # product
    xg = x.to(DEV).requires_grad_(need_x)
    pg = [(W.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for W, b in layers]
    y = ops.mlp(xg, pg, [m.to(DEV) for m in masks] if masks else None, degs=degs if degs else None)
    (y * gy.to(DEV)).sum().backward()
    errs = {"y": rel(y, a)}
    if need_x:
        errs["gx"] = rel(xg.grad, xr.grad)
    bad = []
    for l, ((W, b), (W6, b6)) in enumerate(zip(pg, ps)):
        errs["gW%d" % l], errs["gb%d" % l] = rel(W.grad, W6.grad), rel(b.grad, b6.grad)
        if masks and int(((W.grad.cpu() != 0) & (masks[l] == 0)).sum()):
            bad.append("gW%d non-zero under the mask" % l)
    # fp32 chains against fp64: 1e-5 on outputs of O(1), 1e-4 on gradients (sums over up to 60 000 rows)
    bad += [k for k, v in errs.items() if not v < (2e-5 if k == "y" else 1e-4)]
    desc = "M %5d dims %-28s mask %-10s x.grad %d knife rows %d" % (M, dims, mask_kind, need_x, int(knife.sum()))
    return desc, errs, bad


def walk(n, seed, only=None):
    rng = random.Random(seed)
    out = []
    for case in range(n):
        r = one(case, rng, only)
        if r is not None:
            out.append((case,) + r)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    fails = 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None          # [n] [seed] [case]: that case of the walk alone
    for case, desc, errs, bad in walk(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0, only):
        if only is not None:
            print(errs)
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("case %3d %s worst %s %.1e %s" % (case, desc, worst[0], worst[1], ("FAIL " + ",".join(bad)) if bad else "ok"), flush=True)
        fails += bool(bad)
    print("%d cases, %d failures" % (n, fails))
    sys.exit(1 if fails else 0)
