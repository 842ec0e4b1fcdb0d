"""Round 6, review item 2: the error of an fp32 product formed on the bf16 matrix pipe from exact three-way operand splits
(gnf_gemm_split_bf16: six cross terms, fp32 accumulate) against the error of the fp32-MFMA kernels the product runs today
(gnf_gemm), both measured against an fp64 product of the SAME fp32 operands:
  * the three fc1 GEMMs of the headline step on the operands of a real cfg4 training step (pooled features, fc1 weight, the
    gated cotangent of fc1's output), and on N(0,1) operands of the same shapes;
  * the shapes of tests/fuzz_gemm.py's walk (random M, N, K, operand orders).
Reported per case: max |err| and rms err, both relative to the rms of the fp64 result, for fp32-MFMA and for the split
kernel with 1 / 2 / 3 accumulator classes.     python tools/split_bf16_error.py [n_fuzz] > profiles/r06_split_bf16_error.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops, configs  # noqa: E402
from gnf_hip.abi import ptr, call, stream  # noqa: E402

DEV = "cuda:0"


def f32_gemm(A, sa, B, sb, M, N, K):
    """the fp32-MFMA kernel of the shape: gnf_gemm with its workspace, or -- where that call now dispatches to a split kernel --
    without one (the single-pass kernels gemm_tall_k / gemm_wide_k take none; this is what GNF_TRUE_F32=1 runs)"""
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A, sa, B, sb, C, C.stride(), M, N, K)
    kern = abi.load().gnf_gemm_last_kernel().decode()
    if "split" in kern:
        call("gnf_gemm", ptr(A), sa[0], sa[1], ptr(B), None, sb[0], sb[1], ptr(C), N, 1, None, None, 0, 0, None, 0, 0, 0, M, N, K,
             None, 0, stream())
        kern = abi.load().gnf_gemm_last_kernel().decode()
    return C, kern


_wsbuf = {}


def split_gemm(A, sa, B, sb, M, N, K, classes, splits=1):
    """classes 1..3: the general split kernel; classes 0: the product's choice (a dedicated kernel where the shape has one)"""
    lib = abi.load()
    if splits == 1:
        C = torch.empty(M, N, device=DEV)
        nws = int(lib.gnf_gemm_split_ws_bytes(M, N, K)) if classes == 0 else 0
        ws = _wsbuf.setdefault(nws, torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV))
        call("gnf_gemm_split_bf16", ptr(A), sa[0], sa[1], ptr(B), sb[0], sb[1], ptr(C), N, 1, None, 0, M, N, K, classes, 1, 0,
             abi.rawptr(ws) if nws else None, nws, stream())
        return C
    P = torch.empty(splits, M, N, device=DEV)
    call("gnf_gemm_split_bf16", ptr(A), sa[0], sa[1], ptr(B), sb[0], sb[1], ptr(P), N, 1, None, 0, M, N, K, classes or 3, splits,
         M * N, None, 0, stream())
    C = P[0].clone()
    for z in range(1, splits):                # fixed order, as gemm_reduce_k sums its partials
        C += P[z]
    return C


def ref64(A, sa, B, sb, M, N, K, rows=None):
    """fp64 product of the same fp32 operands (on the device), optionally on a subset of output rows"""
    Am = torch.as_strided(A, (M, K), sa).double()
    Bm = torch.as_strided(B, (K, N), sb).double()
    if rows is not None:
        Am = Am[rows]
    return Am @ Bm


def report(tag, A, sa, B, sb, M, N, K, splits=1, rows=None):
    ref = ref64(A, sa, B, sb, M, N, K, rows)
    scale = ref.pow(2).mean().sqrt().item() or 1.
    out = []
    C, kern = f32_gemm(A, sa, B, sb, M, N, K)
    cands = [("fp32-MFMA (%s)" % kern, C)] + [("split-bf16 x%d acc" % c, split_gemm(A, sa, B, sb, M, N, K, c, splits)) for c in (1, 2, 3)]
    if splits == 1:
        Cd = split_gemm(A, sa, B, sb, M, N, K, 0)
        dk = abi.load().gnf_gemm_split_last_kernel().decode()
        if dk != "gemm_split_k":
            cands.append(("split-bf16 PRODUCT (%s)" % dk, Cd))
    for name, Cx in cands:
        d = (Cx[rows] if rows is not None else Cx).double() - ref
        out.append((name, d.abs().max().item() / scale, d.pow(2).mean().sqrt().item() / scale))
    print("%-34s M=%-6d N=%-5d K=%-6d" % (tag, M, N, K) + "".join("  | %s: max %.2e rms %.2e" % o for o in out), flush=True)
    return out


def cfg4_operands():
    """pooled [78400, 2304], Wfc1 [128, 2304], gated cotangent of fc1's output [78400, 128] of one real cfg4 training step:
    copied out of the fc1 layer's gnf_linear_bwd call (its `g` and `a` device pointers) while the step runs"""
    import ctypes
    torch.manual_seed(0)
    flow = configs.build_cfg4_flow().to(DEV)
    x = configs.pseudo_mnist(torch.Generator().manual_seed(1234), 100, 784).to(DEV)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    got = {}
    orig = ops.call

    def spy(name, *args):
        orig(name, *args)
        if name == "gnf_linear_bwd" and tuple(args[12:15]) == (78400, 128, 2304) and "g" not in got:
            torch.cuda.synchronize()
            for key, a, cols in (("g", args[0], 128), ("pooled", args[2], 2304)):
                t = torch.empty(78400, cols, device=DEV)
                assert hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), a, t.numel() * 4, 3) == 0
                got[key] = t
    ops.call = spy
    try:
        z, ld = flow(x)
        flow.loss(z, ld).backward()
        torch.cuda.synchronize()
    finally:
        ops.call = orig
    W = flow.steps[0].conditioner.embedding_net.fc1.weight.detach().contiguous().clone()
    return got.get("pooled"), W, got.get("g")


def main():
    n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    abi.load()
    worse, prod_worse = [], []
    rows = torch.cat([torch.arange(0, 400), torch.randint(0, 78400, (3200,)), torch.arange(78400 - 400, 78400)]).to(DEV)
    pooled, W, g = cfg4_operands()
    sets = [("N(0,1)", torch.randn(78400, 2304, device=DEV), torch.randn(128, 2304, device=DEV) / 48., torch.randn(78400, 128, device=DEV))]
    if pooled is not None and g is not None:
        sets.insert(0, ("cfg4 step", pooled, W, g))
    else:
        print("# (the fc1 GEMMs of the step did not go through ops.gemm: cfg4 operands not captured)")
    for tag, X, Wf, G in sets:
        M, K, F = X.shape[0], X.shape[1], Wf.shape[0]
        res = [report("fc1 fwd  X W^T   [%s]" % tag, X, (K, 1), Wf, (1, K), M, F, K, rows=rows),
               report("fc1 dX   G W     [%s]" % tag, G, (F, 1), Wf, (K, 1), M, K, F, rows=rows),
               report("fc1 dW   G^T X   [%s]" % tag, G, (1, F), X, (K, 1), F, K, M, splits=14)]
        for r in res:
            worse += [(tag, o[0]) for o in r[1:] if o[1] > r[0][1] or o[2] > r[0][2]]
            prod_worse += [(tag, o[0]) for o in r[1:] if "PRODUCT" in o[0] and (o[1] > r[0][1] or o[2] > r[0][2])]
    gen = torch.Generator().manual_seed(7)
    for i in range(n_fuzz):
        M, N, K = [int(torch.randint(1, hi, (1,), generator=gen)) for hi in (3000, 700, 5000)]
        order = int(torch.randint(0, 4, (1,), generator=gen))
        A = torch.randn(M, K, device=DEV) if order & 1 == 0 else torch.randn(K, M, device=DEV)
        sa = (K, 1) if order & 1 == 0 else (1, M)
        B = torch.randn(K, N, device=DEV) if order & 2 == 0 else torch.randn(N, K, device=DEV)
        sb = (N, 1) if order & 2 == 0 else (1, K)
        r = report("walk %2d (A %s, B %s)" % (i, "m-major" if order & 1 == 0 else "k-major", "k-major" if order & 2 == 0 else "n-major"),
                   A, sa, B, sb, M, N, K)
        worse += [("walk %d" % i, o[0]) for o in r[1:] if o[1] > r[0][1] or o[2] > r[0][2]]
    print("# ADOPTION CRITERION -- cases in which a split kernel THE PRODUCT DISPATCHES TO has a larger max or rms error than the "
          "fp32-MFMA kernel it replaces: %d %r" % (len(prod_worse), prod_worse))
    print("# (the general kernel gemm_split_k is measurement / fallback only: gnf_gemm never dispatches to it; on the random walk "
          "the fp32 kernels split K over workgroups for small outputs, which shortens THEIR summation chains)")
    print("# cases in which any split form has a larger max or rms error than the fp32-MFMA kernel: %d" % len(worse))
    by = {}
    for tag, name in worse:
        by[name] = by.get(name, 0) + 1
    print("#   by form:", by)


if __name__ == "__main__":
    main()
