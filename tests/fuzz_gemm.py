"""Random shapes / layouts / epilogues through gnf_gemm against an fp64 torch product: the dispatcher picks among six kernel
families (gemm_tall_k, gemm_wide_k, gemm_kmajor_k, gemm_vec_k tiles, split-K + reduce, scalar tails) from M, N, K, the strides
and the alignment -- exactly what a random walk varies.   python tests/fuzz_gemm.py [n_cases] [seed]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd")]
from gnf_hip import ops, abi       # noqa: E402
DEV = "cuda:0"
EDGE = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 30, 31, 32, 33, 60, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 784, 1024, 2304]


def dim(rng, big):
    r = rng.random()
    if r < .5:
        return rng.choice(EDGE)
    if r < .85:
        return rng.randint(1, 700)
    return rng.randint(1, big)


def operand(rows, cols, layout, pad, gen):
    """[rows, cols] view with the given memory order ('r' row-major, 'c' column-major) and `pad` extra leading elements"""
    if layout == "r":
        base = torch.randn(rows, cols + pad, generator=gen).to(DEV)
        return base[:, :cols]
    base = torch.randn(cols, rows + pad, generator=gen).to(DEV)
    return base[:, :rows].t()


def one(case, rng):
    shape_kind = rng.random()
    forced = None
    if shape_kind < .12:     # gemm_tall_k's domain and its edges: both operands k-contiguous, 96 < N <= 128, K % 32 == 0, K >= 256
        M, N, K = rng.choice([200, 1000, 5000, 33333, 78400]), rng.choice([96, 97, 100, 112, 127, 128, 129]), \
            rng.choice([224, 256, 288, 512, 1024, 2300, 2304])
        forced = ("r", "c", "r", rng.choice([0, 4, 8, 1]), rng.choice([0, 4, 1]), 0)
        shape_kind = 1.
    elif shape_kind < .2:    # gemm_wide_k: K = 128 exactly, A k-contiguous, B and C n-contiguous, no epilogue
        M, N, K = rng.choice([64, 1000, 7777, 78400]), rng.choice([128, 130, 1000, 2304, 2305]), rng.choice([128, 128, 128, 124, 132])
        forced = ("r", "r", "r", rng.choice([0, 4]), rng.choice([0, 4, 1]), rng.choice([0, 4]))
        shape_kind = 1.
    elif shape_kind < .3:    # gemm_kmajor_k: A m-contiguous, B n-contiguous, M <= 128 a multiple of 4, N >= 512, K >= 16384
        M, N, K = rng.choice([4, 28, 30, 64, 128, 132]), rng.choice([508, 512, 1000, 2304]), rng.choice([16383, 16384, 20000, 78400])
        forced = ("c", "r", "r", rng.choice([0, 4, 2]), rng.choice([0, 4]), 0)
        shape_kind = 1.
    elif shape_kind < .4:    # tall / skinny (the fc1 forward and its gradients, scaled down)
        M, N, K = rng.choice([4096, 6000, 20000, 40000]), rng.choice([16, 30, 64, 128]), rng.choice([32, 64, 128, 400, 2304])
        if rng.random() < .5:
            N, K = K, N
    elif shape_kind < .5:    # long-K weight gradients
        M, N, K = rng.choice([30, 64, 128]), rng.choice([128, 400, 1024, 2304]), rng.choice([4096, 20000, 78400])
    else:
        M, N, K = dim(rng, 5000), dim(rng, 3000), dim(rng, 5000)
    la, lb, lc = rng.choice("rc"), rng.choice("rc"), rng.choice("rrrc")
    pa, pb, pc = rng.choice([0, 0, 1, 3, 4]), rng.choice([0, 0, 1, 3, 4]), rng.choice([0, 0, 0, 1, 4])
    if forced is not None:
        la, lb, lc, pa, pb, pc = forced
    elif shape_kind < .5 and rng.random() < .7:     # the operand orders the dedicated kernels are built for, aligned
        la, lb, lc = rng.choice([("r", "c", "r"), ("r", "r", "r"), ("c", "r", "r")])
        pa, pb, pc = rng.choice([0, 4]), rng.choice([0, 4]), 0
    gen = torch.Generator().manual_seed(case)
    A = operand(M, K, la, pa, gen)
    B = operand(K, N, lb, pb, gen)
    C = operand(M, N, lc, pc, gen)
    use_bias, use_relu = rng.random() < .4, rng.random() < .3
    use_bmask, use_cmask, use_gate = rng.random() < .15, rng.random() < .1, rng.random() < .1
    if forced is not None and rng.random() < .8:    # (the dedicated kernels take bias / ReLU at most)
        use_bmask = use_cmask = use_gate = False
        if forced[0] != "r" or forced[1] != "c":    # wide and k-major: no epilogue at all
            use_bias = use_relu = False
    bias = torch.randn(N, generator=gen).to(DEV) if use_bias else None
    Bmask = (operand(K, N, lb, pb, gen) > -.5).float() if use_bmask else None   # (shares B's strides: same order, same pad)
    if use_bmask:
        Bmask = torch.empty_strided(B.shape, B.stride(), device=DEV).copy_(Bmask) if Bmask.stride() != B.stride() else Bmask
    Cmask = (torch.rand(M, N, generator=gen) < .7).float().to(DEV) if use_cmask else None
    gate = torch.randn(M, N, generator=gen).to(DEV) if use_gate else None
    ops.gemm(A, A.stride(), B, B.stride(), C, C.stride(), M, N, K,
             Bmask=Bmask, bias=bias, Cmask=Cmask, cm_strides=Cmask.stride() if use_cmask else (0, 0),
             gate=gate, g_strides=gate.stride() if use_gate else (0, 0), relu=use_relu)
    kern = abi.load().gnf_gemm_last_kernel().decode()
    ref = A.double() @ (B.double() * (Bmask.double() if use_bmask else 1.))
    mag = A.double().abs() @ (B.double().abs() * (Bmask.double() if use_bmask else 1.))       # bound on the rounding error
    if use_bias:
        ref, mag = ref + bias.double(), mag + bias.double().abs()
    if use_cmask:
        ref = ref * Cmask.double()
    if use_relu:
        ref = torch.relu(ref)
    if use_gate:
        ref = ref * (gate.double() > 0)
    err = (C.double() - ref).abs()
    tol = 2e-6 * mag + 1e-30             # ~sqrt(K) ulps would do; K ulps of the term magnitudes is the safe bound for fp32 chains
    worst = float((err / tol).max())
    desc = "M %6d N %5d K %6d  A%s%d B%s%d C%s%d %s%s%s%s%s %-22s" % (M, N, K, la, pa, lb, pb, lc, pc, "b" if use_bias else "-",
                                                                 "r" if use_relu else "-", "m" if use_bmask else "-",
                                                                 "c" if use_cmask else "-", "g" if use_gate else "-", kern)
    return desc, worst, worst > 1.


def walk(n, seed):
    rng = random.Random(seed)
    out = []
    for case in range(n):
        desc, worst, bad = one(case, rng)
        out.append((case, desc, worst, bad))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fails = 0
    for case, desc, worst, bad in walk(n, seed):
        print("case %3d %s err/tol %.2f %s" % (case, desc, worst, "FAIL" if bad else "ok"), flush=True)
        fails += bad
    print("%d cases, %d failures" % (n, fails))
    sys.exit(1 if fails else 0)
