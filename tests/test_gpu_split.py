"""GPU: fp32 products on the bf16 matrix pipe (gnf_gemm_split.hip, round 6; reference models/MLP.py:44 -- MNISTCNN.fc1 -- and
its autograd).  Every fp32 operand is split exactly into three bf16 numbers and a product is the sum of its six leading cross
terms in an fp32 accumulator.  The adoption criterion of the review: against an fp64 product of the same fp32 operands the
split kernels are AT LEAST as accurate as the fp32-MFMA kernels they replace -- checked here on every shape class the
dispatch sends to them (and at length by tools/split_bf16_error.py -> profiles/r06_split_bf16_error.txt)."""
import pytest
import torch

from gnf_hip import abi, ops
from gnf_hip.abi import ptr, call, stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _f32(A, sa, B, sb, M, N, K, bias=None, relu=False, ws=False):
    """gnf_gemm: ws=False -> a workspace of exactly gnf_gemm_f32_ws_bytes (the fp32-MFMA kernels with their split-K partials: too
    small for the split-bf16 dispatch; what GNF_TRUE_F32=1 runs); ws=True -> the full gnf_gemm_ws_bytes (the product's call)"""
    C = torch.empty(M, N, device=DEV)
    lib = abi.load()
    nws = int(lib.gnf_gemm_ws_bytes(M, N, K)) if ws else int(lib.gnf_gemm_f32_ws_bytes(M, N, K))
    ws = ws or nws > 0
    w = torch.empty(max(nws // 4, 1), device=DEV) if ws else None
    call("gnf_gemm", ptr(A), sa[0], sa[1], ptr(B), None, sb[0], sb[1], ptr(C), N, 1, ptr(bias), None, 0, 0, None, 0, 0,
         1 if relu else 0, M, N, K, ptr(w), nws, stream())
    return C, lib.gnf_gemm_last_kernel().decode()


def _split(A, sa, B, sb, M, N, K, bias=None, relu=False, classes=0):
    C = torch.empty(M, N, device=DEV)
    lib = abi.load()
    nws = int(lib.gnf_gemm_split_ws_bytes(M, N, K))
    w = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
    call("gnf_gemm_split_bf16", ptr(A), sa[0], sa[1], ptr(B), sb[0], sb[1], ptr(C), N, 1, ptr(bias), 1 if relu else 0, M, N, K,
         classes, 1, 0, abi.rawptr(w), nws, stream())
    return C, lib.gnf_gemm_split_last_kernel().decode()


def _errs(C, ref):
    d = C.double() - ref
    sc = ref.pow(2).mean().sqrt().item() or 1.
    return d.abs().max().item() / sc, d.pow(2).mean().sqrt().item() / sc


@pytest.mark.parametrize("M,N,K,bias,relu", [(78400, 128, 2304, True, True),      # cfg4 fc1 forward
                                             (10240, 128, 256, False, False),    # the smallest eligible block count, short K
                                             (10301, 100, 1024, True, False),    # ragged last block, N < 128 (zero-padded planes)
                                             (20000, 65, 384, False, True)])
def test_split_tall_kernel_vs_fp64_and_fp32_mfma(M, N, K, bias, relu):
    torch.manual_seed(M + N)
    X, W = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** .5
    b = torch.randn(N, device=DEV) if bias else None
    rows = torch.randint(0, M, (3000,), device=DEV)
    rows[:200] = torch.arange(M - 200, M, device=DEV)                    # the ragged tail is in the sample
    ref = X[rows].double() @ W.double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = torch.relu(ref)
    Cs, ks = _split(X, (K, 1), W, (1, K), M, N, K, b, relu)
    Cf, kf = _f32(X, (K, 1), W, (1, K), M, N, K, b, relu)
    assert ks == "gemm_split_tall_k", ks
    es, ef = _errs(Cs[rows], ref), _errs(Cf[rows], ref)
    assert es[0] <= ef[0] and es[1] <= ef[1], (es, ef, kf)
    assert es[0] < 1e-5 and es[1] < 1e-6, es
    # deterministic
    Cs2, _ = _split(X, (K, 1), W, (1, K), M, N, K, b, relu)
    assert torch.equal(Cs, Cs2)


@pytest.mark.parametrize("M,N", [(78400, 2304), (10240, 512), (10333, 1280)])
def test_split_wide_kernel_vs_fp64_and_fp32_mfma(M, N):
    K = 128
    torch.manual_seed(M + N)
    G, W = torch.randn(M, K, device=DEV), torch.randn(K, N, device=DEV) / K ** .5
    rows = torch.randint(0, M, (3000,), device=DEV)
    rows[:200] = torch.arange(M - 200, M, device=DEV)
    ref = G[rows].double() @ W.double()
    Cs, ks = _split(G, (K, 1), W, (N, 1), M, N, K)
    Cf, kf = _f32(G, (K, 1), W, (N, 1), M, N, K)
    assert ks == "gemm_split_wide_k", ks
    es, ef = _errs(Cs[rows], ref), _errs(Cf[rows], ref)
    assert es[0] <= ef[0] and es[1] <= ef[1], (es, ef, kf)
    assert es[0] < 4e-6 and es[1] < 3e-7, es
    Cs2, _ = _split(G, (K, 1), W, (N, 1), M, N, K)
    assert torch.equal(Cs, Cs2)


@pytest.mark.parametrize("M,N,K", [(128, 2304, 78400),        # cfg4 fc1 weight gradient
                                   (64, 512, 20001),          # M < 128 (zero-padded planes), K neither a multiple of 64 nor of the ranges
                                   (112, 640, 33000), (128, 1280, 16384)])
def test_split_kmajor_kernel_vs_fp64_and_fp32_mfma(M, N, K):
    torch.manual_seed(M + N + K)
    G, X = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
    ref = G.double().t() @ X.double()
    Cs, ks = _split(G, (1, M), X, (N, 1), M, N, K)
    Cf, kf = _f32(G, (1, M), X, (N, 1), M, N, K)
    assert ks == "gemm_split_kmajor_k" and kf == "gemm_kmajor_k", (ks, kf)
    es, ef = _errs(Cs, ref), _errs(Cf, ref)
    assert es[0] <= ef[0] and es[1] <= ef[1], (es, ef)
    assert es[0] < 6e-6 and es[1] < 1e-6, es
    Cs2, _ = _split(G, (1, M), X, (N, 1), M, N, K)
    assert torch.equal(Cs, Cs2)
    # strided rows of both operands and of C (views into wider buffers)
    Gw, Xw = torch.randn(K, M + 4, device=DEV), torch.randn(K, N + 8, device=DEV)
    Cw = torch.zeros(M, N + 4, device=DEV)
    lib = abi.load()
    nws = int(lib.gnf_gemm_split_ws_bytes(M, N, K))
    w = torch.empty(nws, dtype=torch.uint8, device=DEV)
    call("gnf_gemm_split_bf16", ptr(Gw), 1, M + 4, ptr(Xw), N + 8, 1, ptr(Cw), N + 4, 1, None, 0, M, N, K, 0, 1, 0, abi.rawptr(w), nws,
         stream())
    assert lib.gnf_gemm_split_last_kernel().decode() == "gemm_split_kmajor_k"
    refw = Gw[:, :M].double().t() @ Xw[:, :N].double()
    assert _errs(Cw[:, :N], refw)[0] < 6e-6 and float(Cw[:, N:].abs().max()) == 0.


def test_gnf_gemm_routes_the_fc1_shapes_to_the_split_kernels_and_honours_the_switch():
    """gnf_gemm with its workspace: the fc1 forward / data-gradient shapes run on the split kernels (the product path of
    MLPFn -> gnf_linear_* -> gnf_gemm); without a workspace, with an epilogue the split kernels do not have, or for a shape
    outside their domain, the fp32-MFMA kernels run.  GNF_TRUE_F32=1 (read once per process) turns the dispatch off: checked
    in a child process."""
    import os
    import subprocess
    import sys
    lib = abi.load()
    assert lib.gnf_gemm_split_enabled() == (0 if os.environ.get("GNF_TRUE_F32") == "1" else 1)
    if not lib.gnf_gemm_split_enabled():
        pytest.skip("GNF_TRUE_F32=1")
    M, F, K = 20480, 128, 512
    X, W, G = torch.randn(M, K, device=DEV), torch.randn(F, K, device=DEV), torch.randn(M, F, device=DEV)
    _, k1 = _f32(X, (K, 1), W, (1, K), M, F, K, ws=True)
    assert k1 == "gemm_split_tall_k", k1
    _, k2 = _f32(G, (F, 1), W, (K, 1), M, K, F, ws=True)
    assert k2 == "gemm_split_wide_k", k2
    _, k3 = _f32(X, (K, 1), W, (1, K), M, F, K, ws=False)
    assert "split" not in k3, k3
    Gk, Xk = torch.randn(20000, 128, device=DEV), torch.randn(20000, 1024, device=DEV)
    _, k5 = _f32(Gk, (1, 128), Xk, (1024, 1), 128, 1024, 20000, ws=True)
    assert k5 == "gemm_split_kmajor_k", k5
    _, k4 = _f32(X[:4096], (K, 1), W, (1, K), 4096, F, K, ws=True)        # too few rows for the 160-row blocks
    assert "split" not in k4, k4
    code = ("import sys; sys.path[:0] = %r; import torch; from gnf_hip import abi; "
            "print(abi.load().gnf_gemm_split_enabled())" % ([p for p in sys.path if p],))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GNF_TRUE_F32="1"), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("0"), (out.stdout, out.stderr[-500:])


@pytest.mark.parametrize("seed", range(6))
def test_split_general_kernel_random_shapes_vs_fp64(seed):
    """the general split kernel (any strides, any M, N, K; three accumulator classes) against an fp64 product: the error of an
    fp32 product rounded once per K-slab of 32"""
    g = torch.Generator().manual_seed(100 + seed)
    M, N, K = [int(torch.randint(1, hi, (1,), generator=g)) for hi in (700, 300, 1500)]
    order = seed & 3
    A = torch.randn(M, K, device=DEV) if order & 1 == 0 else torch.randn(K, M, device=DEV)
    sa = (K, 1) if order & 1 == 0 else (1, M)
    B = torch.randn(K, N, device=DEV) if order & 2 == 0 else torch.randn(N, K, device=DEV)
    sb = (N, 1) if order & 2 == 0 else (1, K)
    ref = torch.as_strided(A, (M, K), sa).double() @ torch.as_strided(B, (K, N), sb).double()
    C, k = _split(A, sa, B, sb, M, N, K, classes=3)
    assert k == "gemm_split_k"
    e = _errs(C, ref)
    assert e[0] < 8e-6 and e[1] < 6e-7, (M, N, K, e)


def test_mnistcnn_fc1_runs_on_the_split_kernels_in_the_product_path():
    """MLPFn on the fc1 shape of the headline model: forward, data gradient and weight gradient dispatch to the split kernels,
    values and gradients against an fp64 autograd at the usual tolerances."""
    lib = abi.load()
    if not lib.gnf_gemm_split_enabled():
        pytest.skip("GNF_TRUE_F32=1")
    torch.manual_seed(0)
    M, K, F = 17920, 2304, 128                           # (>= 16 384 rows: the weight gradient's long K)
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    W = (torch.randn(F, K, device=DEV) / 48.).requires_grad_(True)
    b = torch.randn(F, device=DEV, requires_grad=True)
    seen = []
    orig = ops.call

    def spy(name, *args):
        orig(name, *args)
        if name.startswith("gnf_linear"):
            seen.append(lib.gnf_gemm_last_kernel().decode())
            seen.append(lib.gnf_gemm_split_last_kernel().decode())
    ops.call = spy
    try:
        y = ops.mlp(x, [(W, b)])
        gy = torch.randn_like(y)
        y.backward(gy)
    finally:
        ops.call = orig
    assert "gemm_split_tall_k" in seen and "gemm_split_wide_k" in seen, seen     # (gnf_linear_bwd runs dW, then dX)
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    y64 = x64 @ W64.t() + b64
    y64.backward(gy.double())
    for got, ref, name in ((y, y64, "y"), (x.grad, x64.grad, "dx"), (W.grad, W64.grad, "dW"), (b.grad, b64.grad, "db")):
        d = (got.double() - ref.detach()).abs().max().item()
        assert d <= 1e-5 * ref.detach().abs().max().item(), (name, d)


def test_split_kernels_inside_a_captured_graph():
    """the three split kernels (pack + main + reduce launches, workspace from the caching allocator) captured into a hipGraph and
    replayed on new inputs: same bits as the eager calls (GraphedStep captures whole optimisation steps that contain them)"""
    lib = abi.load()
    if not lib.gnf_gemm_split_enabled():
        pytest.skip("GNF_TRUE_F32=1")
    torch.manual_seed(3)
    M, K, F = 20480, 512, 128
    x = torch.randn(M, K, device=DEV)
    W = (torch.randn(F, K, device=DEV) / 20.).requires_grad_(True)
    b = torch.randn(F, device=DEV, requires_grad=True)
    xs = x.clone().requires_grad_(True)

    def step():
        W.grad = b.grad = xs.grad = None
        y = ops.mlp(xs, [(W, b)])
        y.square().sum().backward()
        return y.detach(), xs.grad, W.grad, b.grad
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    x2 = torch.randn(M, K, device=DEV)
    with torch.no_grad():
        xs.copy_(x2)
    g.replay()
    got = [t.clone() for t in outs]
    ref = [t.clone() for t in step()]
    for a, r, name in zip(got, ref, ("y", "dx", "dW", "db")):
        assert torch.equal(a, r), name
    x64, W64, b64 = x2.double().requires_grad_(True), W.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    (x64 @ W64.t() + b64).square().sum().backward()
    assert (got[2].double() - W64.grad).abs().max() <= 1e-5 * W64.grad.abs().max()
    assert (got[1].double() - x64.grad).abs().max() <= 1e-5 * x64.grad.abs().max()


def test_training_trajectory_matches_the_true_f32_build():
    """end to end: five Adam steps of the cfg4 flow at B = 16 (12 544 masked copies: the fc1 products are on the split kernels)
    in a child process with the default library and in one with GNF_TRUE_F32=1 -- same seeds, same Philox gate noise: the loss
    of every step agrees to 1e-5 relative and the parameters after the steps to 1e-6 of their L1 norm"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import split_trajectory as T
    if not abi.load().gnf_gemm_split_enabled():
        pytest.skip("GNF_TRUE_F32=1")
    a, fa = T.run(5, 16, False)
    b, fb = T.run(5, 16, True)
    assert len(a) == len(b) == 5
    for u, v in zip(a, b):
        assert abs(u - v) <= 1e-5 * abs(v), (a, b)
    assert abs(fa[1] - fb[1]) <= 1e-6 * fb[1], (fa, fb)
