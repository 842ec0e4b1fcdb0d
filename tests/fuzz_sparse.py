"""Random row subsets / batch sizes / window gates through the sparse masked-image front (gnf_mnistcnn_sparse.hip: crop kernel in
both forms, grouped fc1 GEMM + fc2, and the one-launch fc1 + ReLU + fc2 kernel of the held-table path) against the oracle's
dense MNISTCNN on the masked copies (MLP.py:36-48 over DAGConditioner.py:142-153).   python tests/fuzz_sparse.py [n] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd"), os.path.join(ROOT, "tests")]
from oracle import gnf_oracle as O      # noqa: E402
from gnf_hip import ops                  # noqa: E402
from models.MLP import MNISTCNN          # noqa: E402
DEV = "cuda:0"


def window_gate(gen, density):
    r = torch.arange(28).repeat_interleave(28)
    c = torch.arange(28).repeat(28)
    win = ((r[:, None] - r[None, :]).abs() <= 2) & ((c[:, None] - c[None, :]).abs() <= 2)
    return (win & (torch.rand(784, 784, generator=gen) < density)).float() * (torch.rand(784, 784, generator=gen) + .2)


def one(case, rng):
    gen = torch.Generator().manual_seed(case)
    B = rng.choice([1, 2, 3, 5, 16, 17, 33, 64, 100, 130])
    nrows = rng.choice([1, 1, 2, 3, 7, 8, 9, 16, 28, 50, 200]) if B <= 33 else rng.choice([1, 2, 4, 8, 9, 20])
    kind = rng.random()
    if kind < .3:                                           # one crop origin: every copy in the same group
        base = rng.randrange(784)
        r0, c0 = base // 28, base % 28
        pool = [r * 28 + c for r in range(max(0, r0 - 1), min(28, r0 + 2)) for c in range(max(0, c0 - 1), min(28, c0 + 2))]
        rows = rng.sample(pool, min(nrows, len(pool)))
    elif kind < .5:                                         # a diagonal of the level schedule: c + 3 r = const
        k = rng.randrange(109)
        rows = [r * 28 + (k - 3 * r) for r in range(28) if 0 <= k - 3 * r < 28]
    else:
        rows = rng.sample(range(784), nrows)
    out_d = rng.choice([30, 30, 30, 2, 1, 32])
    torch.manual_seed(case)
    net = MNISTCNN(out_d=out_d).to(DEV)
    P = window_gate(gen, rng.choice([.2, .5, 1.])).to(DEV)
    x = torch.randn(B, 784, generator=gen).to(DEV)
    sr = ops.SparseRows(rows, B, torch.device(DEV))
    with torch.no_grad():
        plain = net.sparse_rows(x, P, sr)                   # tables built per call, grouped GEMM + fc2
        with net.hold_prepared():
            held = net.sparse_rows(x, P, sr)                # held tables: the one-launch fc kernel where it applies
    ps = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    e = (x.cpu().unsqueeze(1) * P.cpu()[rows].unsqueeze(0)).reshape(B * len(rows), 784)
    want = O.mnistcnn_forward(e, ps).view(B, len(rows), out_d)
    def worst(a):
        d = (a.cpu().double() - want.double()).abs() - (1e-6 + 1e-5 * want.double().abs())
        return float(d.max())
    w1, w2 = worst(plain), worst(held)
    desc = "B %3d rows %3d (%d origins) out_d %2d" % (B, len(rows), int((sr.groups.view(64, 2)[:, 1] > 0).sum()), out_d)
    return desc, max(w1, w2), (w1 > 0) or (w2 > 0)


def walk(n, seed):
    rng = random.Random(seed)
    return [(case,) + one(case, rng) for case in range(n)]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    fails = 0
    for case, desc, w, bad in walk(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0):
        print("case %3d %s  excess over atol 1e-6 + rtol 1e-5: %.2e %s" % (case, desc, w, "FAIL" if bad else "ok"), flush=True)
        fails += bad
    print("%d cases, %d failures" % (n, fails))
    sys.exit(1 if fails else 0)
