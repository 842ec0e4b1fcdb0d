"""Random row subsets / batch sizes through the TRAINING path of the sparse front (frozen deterministic gate: crop forward with the
argmax record, grouped fc1 GEMMs, the two-wavefront crop backward, closed-form background terms) against the dense kernels'
gradients on the same masked copies (DAGConditioner.py:142-153 -> MLP.py:36-48 and their autograd).  Masked copies holding a
knife-edge pool window get a zero cotangent when a first comparison fails.   python tests/fuzz_sparse_grad.py [n] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd"), os.path.join(ROOT, "tests")]
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def grads(cond, x, r, P, gh, sparse):
    cond.sparse_front = sparse
    for p in cond.embedding_net.parameters():
        p.grad = None
    (cond.forward_rows(x, r, P) * gh).sum().backward()
    return {k: p.grad.clone() for k, p in cond.embedding_net.named_parameters()}


def one(case, rng):
    from test_gpu_parity import _windowed_conditioner, _knife_edge_windows
    cond = _windowed_conditioner(1000 + case, True)
    B = rng.choice([1, 2, 3, 5, 9, 17])
    n = rng.choice([1, 2, 3, 7, 16, 40, 120])
    kind = rng.random()
    if kind < .3:
        k = rng.randrange(109)
        rows = [r * 28 + (k - 3 * r) for r in range(28) if 0 <= k - 3 * r < 28]
    else:
        rows = sorted(rng.sample(range(784), n))
    g = torch.Generator().manual_seed(case)
    x = torch.rand(B, 784, generator=g).to(DEV)
    r = torch.tensor(rows, device=DEV)
    gh = torch.randn(B, len(rows), 30, generator=g).to(DEV)
    with torch.no_grad():
        P = cond.deterministic_importance()
    gs, gd = grads(cond, x, r, P, gh, True), grads(cond, x, r, P, gh, False)
    errs = {k: rel(gs[k], gd[k]) for k in gs}
    bad = [k for k, v in errs.items() if not v < 1e-4]
    arb = 0
    if bad:
        flips, knife = _knife_edge_windows(cond, x.cpu(), with_images=True)      # indices b * 784 + i of the masked copies
        knife = set(int(v) for v in knife.tolist())
        # ... and the copies with a ReLU on the knife edge (conv1, or fc1 -- which the sparse front evaluates as background +
        # block, another summation order than the dense 2304-wide product), found in fp64 on the copies of this case
        from conftest import conv_front_knife_images, EPS32
        import torch.nn.functional as F
        net = cond.embedding_net
        W1, b1, W2, b2, Wf, bf = [t.detach().cpu().double() for t in (net.conv1.weight, net.conv1.bias, net.conv2.weight,
                                                                      net.conv2.bias, net.fc1.weight, net.fc1.bias)]
        e = (x.cpu().unsqueeze(1) * P.detach().cpu()[rows].unsqueeze(0)).reshape(B * len(rows), 784)
        kimg, _, _ = conv_front_knife_images(e, W1, b1, W2, b2)
        pooled = F.max_pool2d(F.conv2d(torch.relu(F.conv2d(e.double().view(-1, 1, 28, 28), W1, b1)), W2, b2), 2).flatten(1)
        pre = pooled @ Wf.t() + bf
        mag = pooled.abs() @ Wf.abs().t() + bf.abs()
        kfc = ((pre.abs() < 16 * EPS32 * mag) & (pre != 0)).any(1)
        for idx in (kimg | kfc).nonzero().flatten().tolist():
            b, k_ = divmod(idx, len(rows))
            knife.add(b * 784 + rows[k_])
        for b in range(B):
            for k_, i in enumerate(rows):
                if b * 784 + i in knife:
                    gh[b, k_] = 0.
                    arb += 1
        gs, gd = grads(cond, x, r, P, gh, True), grads(cond, x, r, P, gh, False)
        errs = {k: rel(gs[k], gd[k]) for k in gs}
        bad = [k for k, v in errs.items() if not v < 1e-4]
    return "B %2d rows %3d%s" % (B, len(rows), " (%d knife copies zeroed)" % arb if arb else ""), errs, bad


def walk(n, seed):
    rng = random.Random(seed)
    return [(case,) + one(case, rng) for case in range(n)]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    fails = 0
    for case, desc, errs, bad in walk(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0):
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("case %3d %s worst %s %.1e %s" % (case, desc, worst[0], worst[1], ("FAIL " + ",".join(bad)) if bad else "ok"), flush=True)
        fails += bool(bad)
    print("%d cases, %d failures" % (n, fails))
    sys.exit(1 if fails else 0)
