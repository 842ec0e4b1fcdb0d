"""Per-kernel roofline table: each C-ABI entry point at representative sizes, timed with HIP events
(median of N launches), against its algorithmic bytes / flops.
Usage: python tools/bench_kernels.py [--json out.json]"""
import json
import sys

import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops  # noqa: E402
from _warm import warm_gpu  # noqa: E402
from models import MonotonicNormalizer  # noqa: E402

DEV = "cuda:0"
HBM_PEAK, F32_PEAK, BF16_PEAK = 8000., 157.3, 2500.   # GB/s (spec), TFLOP/s (fp32 MFMA), TFLOP/s (dense bf16 MFMA)


def timeit(fn, n=20, warm=3):
    warm_gpu(.15)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def time_entry(name, fn, n=20, warm=3):
    """median-free mean of the HIP events gnf_hip.abi records around the NAMED C-ABI entry point (on its launch stream):
    the kernel(s) of that entry point alone, without autograd bookkeeping, output allocation or other launches"""
    warm_gpu(.15)
    for _ in range(warm):
        fn()
    abi.profile_enable((name,))
    for _ in range(n):
        fn()
    return abi.profile_collect()[name]


rows = []


def hbm(name, shape, ms, nbytes):
    gbs = nbytes / ms / 1e6
    rows.append({"kernel": name, "shape": shape, "ms": round(ms, 4), "bound": "hbm", "achieved_GBps": round(gbs, 1),
                 "frac_of_8TBps": round(gbs / HBM_PEAK, 3)})


def b2b(fn, n=50, reps=9):
    """milliseconds per launch with `n` launches per HIP-event pair (launches pipeline; an event pair around ONE launch on an
    otherwise idle stream includes ~6 us of launch + ramp that an EMPTY grid of the same shape also takes)"""
    warm_gpu(.15)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    ts.sort()
    return ts[len(ts) // 2]


def mfma(name, shape, ms, flops, direct=None):
    """flops = what the kernel must execute in the algorithm it implements; direct = the direct-convolution count of
    SURVEY.md 8(d) for the Winograd kernels (reported beside it, never as the fraction: it exceeds 1)"""
    tf = flops / ms / 1e9
    rows.append({"kernel": name, "shape": shape, "ms": round(ms, 4), "bound": "mfma", "achieved_TFLOPs": round(tf, 1),
                 "frac_of_157TF": round(tf / F32_PEAK, 3)})
    assert 0. < rows[-1]["frac_of_157TF"] <= 1., rows[-1]
    if direct is not None:
        rows[-1]["flop_basis"] = "executed: Winograd F(2x2,3x3)-domain products (conv2-sized parts x 4/9; da1 on its 13x13 tile grid)"
        rows[-1]["direct_conv_equivalent_TFLOPs"] = round(direct / ms / 1e9, 1)
        rows[-1]["frac_direct_conv_equivalent"] = round(direct / ms / 1e9 / F32_PEAK, 3)


def main():
    torch.manual_seed(0)
    global lib0
    lib0 = abi.load()
    # ---- Affine normalizer (16 B/elem fwd, 28 B/elem bwd)
    for B, d in [(100, 784), (50000, 63), (1000000, 63), (4000000, 63)]:     # the last one: 1 GB per array, beyond the 256 MB Infinity Cache
        x = torch.randn(B, d, device=DEV, requires_grad=True)
        h = torch.randn(B, d, 2, device=DEV, requires_grad=True)
        with torch.no_grad():
            hbm("affine_fwd(z,logdet)", [B, d], time_entry("gnf_affine_fwd", lambda: ops.AffineFn.apply(x, h, False, False)), 16. * B * d + 4 * B)
            hbm("affine_fwd(z,logdet,logN)", [B, d], time_entry("gnf_affine_fwd", lambda: ops.AffineFn.apply(x, h, False, False, True)), 16. * B * d + 8 * B)
        z, jac, ld, _ = ops.AffineFn.apply(x, h)
        gz, gl = torch.randn_like(z), torch.randn_like(ld)
        hbm("affine_bwd", [B, d], time_entry("gnf_affine_bwd", lambda: torch.autograd.grad((z, ld), (x, h), (gz, gl), retain_graph=True)),
            28. * B * d + 4 * B)
        hbm("affine_bwd (through autograd)", [B, d], timeit(lambda: torch.autograd.grad((z, ld), (x, h), (gz, gl), retain_graph=True)),
            28. * B * d + 4 * B)
        if d <= 64 and B * d >= (1 << 16):
            # the C-ABI entry points back to back on preallocated outputs, next to the floors of their launch shape (review of round
            # 5, item 7): an EMPTY grid of the same shape and the STREAM copy kernel moving the same bytes, all timed the same way
            P, st = abi.ptr, abi.stream
            zz, ll = torch.empty(B, d, device=DEV), torch.empty(B, device=DEV)
            gx_, gh_ = torch.empty(B, d, device=DEV), torch.empty(B, d, 2, device=DEV)
            xd, hd = x.detach(), h.detach()
            U = 4 if B * d >= (1 << 24) else 1
            grid = min((B + 16 * U - 1) // (16 * U), 4096)
            fb, bb = 16. * B * d + 4 * B, 28. * B * d + 4 * B
            src = torch.empty(int(bb) // 8 // 4 * 4 + 4, device=DEV)
            dst = torch.empty_like(src)
            for nm, fn, nbytes in (
                    ("gnf_affine_fwd, back to back", lambda: abi.call("gnf_affine_fwd", P(xd), P(hd), 2 * d, 2, 1, P(zz), None, P(ll), None, 0, B, d, st()), fb),
                    ("gnf_affine_bwd, back to back", lambda: abi.call("gnf_affine_bwd", P(xd), P(hd), 2 * d, 2, 1, P(gz), None, P(gl), None, P(gx_), P(gh_), 2 * d, 2, 1, B, d, st()), bb),
                    ("floor: STREAM copy of the forward's bytes, back to back", lambda: lib0.gnf_probe_copy(P(dst), P(src), int(fb) // 8 // 4 * 4, st()), 8. * (int(fb) // 8 // 4 * 4)),
                    ("floor: STREAM copy of the backward's bytes, back to back", lambda: lib0.gnf_probe_copy(P(dst), P(src), int(bb) // 8 // 4 * 4, st()), 8. * (int(bb) // 8 // 4 * 4))):
                hbm(nm, [B, d], b2b(fn), nbytes)
            rows.append({"kernel": "floor: EMPTY grid of the Affine launch shape (%d x 256)" % grid, "shape": [B, d], "bound": "launch",
                         "ms": round(timeit(lambda: lib0.gnf_probe_empty(grid, 256, st())), 4),
                         "ms_back_to_back": round(b2b(lambda: lib0.gnf_probe_empty(grid, 256, st())), 4)})
        with torch.no_grad():
            hbm("normal_logdensity_fwd", [B, d], time_entry("gnf_normal_logdensity_fwd", lambda: ops.NormalLogDensityFn.apply(x)), 4. * B * d)
            jj = torch.rand(B, d, device=DEV) + .1
            hbm("nll_reduce_fwd(logdet,logN)", [B, d], time_entry("gnf_nll_reduce_fwd", lambda: ops.NllReduceFn.apply(x, jj)), 8. * B * d)
    # ---- DAG gate (writes 4 B per (b,i,j))
    for B, d in [(100, 784), (10000, 6)]:
        x = torch.randn(B, d, device=DEV)
        A = torch.rand(d, d, device=DEV)
        with torch.no_grad():
            f = lambda: ops.DagGateFn.apply(x, A, ops.IMP_SOFT, ops.GATE_GUMBEL, 0., 1., False, None, None, 1, 1)
            hbm("dag_gate_fwd(gumbel)", [B, d, d], timeit(f), 4. * B * d * d)
    # ---- GEMM shapes on the measured configurations, THROUGH THE STRIDES THE PRODUCT USES (ops.MLPFn -> gnf_linear_* -> gnf_gemm):
    #      forward  y = a W^T        A = a [M,K] row-major, B[k][n] = W[n][k] (k-contiguous: strides (1, K))
    #      dX       ga = g W         A = g [M,K] row-major, B = W [K,N] row-major (n-contiguous: strides (N, 1))
    #      dW       gW = g^T a       A[m][k] = g[k][m] (k-major: strides (1, M)), B = a [K,N] row-major (strides (N, 1))
    #      every row names the kernel family that ran (gnf_gemm_last_kernel)
    lib = abi.load()
    for M, N, K, what, layout in [(78400, 128, 2304, "cfg4 fc1 fwd", "fwd"), (78400, 2304, 128, "cfg4 fc1 dX", "dx"),
                                  (128, 2304, 78400, "cfg4 fc1 dW (split-K)", "dw"), (100, 1024, 1024, "cfg3 MADE hidden", "fwd"),
                                  (50000, 630, 630, "cfg5 MADE hidden", "fwd"), (4096, 4096, 4096, "square", "fwd")]:
        C = torch.empty(M, N, device=DEV)
        if layout == "fwd":
            Am, Bm = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
            sa, sb = (K, 1), (1, K)
        elif layout == "dx":
            Am, Bm = torch.randn(M, K, device=DEV), torch.randn(K, N, device=DEV)
            sa, sb = (K, 1), (N, 1)
        else:
            Am, Bm = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
            sa, sb = (1, M), (N, 1)
        ms = timeit(lambda: ops.gemm(Am, sa, Bm, sb, C, (N, 1), M, N, K))
        ran = lib.gnf_gemm_last_kernel().decode()
        if ran.startswith("gemm_split"):
            # round 6: the fc1 trio on the bf16 matrix pipe -- six v_mfma_f32_16x16x32_bf16 terms per fp32 product: priced at
            # what it executes (6 x 2 M N K) against the dense bf16 peak; the fp32-equivalent rate beside it.  These kernels run at
            # the board's 1400 W cap (1.9-2.05 GHz, profiles/r06_split_bf16_ab.txt); the time includes their pack / reduce launches
            tf = 6. * 2. * M * N * K / ms / 1e9
            rows.append({"kernel": "gemm " + what, "shape": [M, N, K], "ms": round(ms, 4), "bound": "mfma (bf16, 6 terms per fp32 product)",
                         "achieved_TFLOPs_bf16": round(tf, 1), "frac_of_2500TF_bf16": round(tf / BF16_PEAK, 3),
                         "fp32_equivalent_TFLOPs": round(2. * M * N * K / ms / 1e9, 1)})
            assert 0. < rows[-1]["frac_of_2500TF_bf16"] <= 1., rows[-1]
        else:
            mfma("gemm " + what, [M, N, K], ms, 2. * M * N * K)
        rows[-1]["kernel_ran"] = ran
        rows[-1]["operand_strides"] = {"A": list(sa), "B": list(sb)}
    # ---- small-batch masked Linear (cfg3's MADE hidden layer, mask as degree vectors): weights streamed once (4 N K bytes)
    M, N, K = 100, 1024, 1024
    xl = torch.randn(M, K, device=DEV, requires_grad=True)
    Wl, bl = (torch.randn(N, K, device=DEV) / 32).requires_grad_(True), torch.zeros(N, device=DEV, requires_grad=True)
    W2l, b2l = (torch.randn(N, N, device=DEV) / 32).requires_grad_(True), torch.zeros(N, device=DEV, requires_grad=True)
    dg = (783 - torch.arange(N, device=DEV) % 784).float()
    mk = (dg[None, :] <= dg[:, None]).float()

    def lin_step():
        for t in (xl, Wl, bl, W2l, b2l):
            t.grad = None
        ops.mlp(xl, [(Wl, bl), (W2l, b2l)], [mk, mk], degs=[(dg, dg, False), (dg, dg, False)]).sum().backward()
    for entry, nw in (("gnf_linear_fwd", 1), ("gnf_linear_bwd", 2)):       # bwd: both gradients in one launch, weights read twice
        ms = time_entry(entry, lin_step)
        rows.append({"kernel": entry + " (MADE mask as degrees)", "shape": [M, N, K], "ms": round(ms, 4), "bound": "hbm (weights once per product)",
                     "achieved_GBps": round(4. * nw * N * K / ms / 1e6, 1), "frac_of_8TBps": round(4. * nw * N * K / ms / 1e6 / HBM_PEAK, 3),
                     "achieved_TFLOPs": round(2. * nw * M * N * K / ms / 1e9, 1),
                     "note": "HIP events around ONE launch on an idle stream: includes ~3 us of dispatch; rocprofv3 kernel durations in profiles/r03_linear_kernels.txt"})
    # ---- tall batch, narrow output (gnf_linear_tall.hip): fc2 of the cfg4 embedding net, a DAGMLP hidden layer of cfg2
    for M, N, K, what in [(78400, 30, 128, "cfg4 MNISTCNN.fc2"), (60000, 60, 60, "cfg2 DAGMLP hidden")]:
        xt = torch.relu(torch.randn(M, K, device=DEV)).requires_grad_(True)
        Wt, bt = (torch.randn(N, K, device=DEV) / K ** .5).requires_grad_(True), torch.zeros(N, device=DEV, requires_grad=True)

        def tall_step():
            for t in (xt, Wt, bt):
                t.grad = None
            ops.mlp(xt, [(Wt, bt)], relu_in=True).sum().backward()
        for entry, nbytes in (("gnf_linear_fwd", 4. * M * (K + N)), ("gnf_linear_bwd", 4. * M * (N + 2 * K))):
            hbm(entry + " " + what, [M, N, K], time_entry(entry, tall_step), nbytes)
    # ---- Monotonic quadrature, forward: 2*M*(S+2) flop per element
    for (B, d, c, hid, tag) in [(100, 784, 30, [50, 50, 50], "cfg4"), (10000, 6, 30, [100, 100, 100], "cfg2"),
                                (50000, 63, 30, [150, 150, 150], "cfg5")]:
        S = 20
        norm = MonotonicNormalizer(hid, c, nb_steps=S).to(DEV)
        x, h = torch.randn(B, d, device=DEV), torch.randn(B, d, c, device=DEV)
        dims = [1 + c] + hid + [1]
        macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        split_fwd = hid[0] > 96 and lib0.gnf_gemm_split_enabled()
        split_bwd = hid[0] > 112 and lib0.gnf_gemm_split_enabled()

        def mono_row(name, ms, passes, split):
            """fp32 kernels: algorithmic flop against the fp32 MFMA peak.  Split kernels (round 6): the hidden->hidden products run
            as six v_mfma_f32_16x16x32_bf16 terms on padded tiles -- priced at what they execute against the dense bf16 peak, the
            fp32-equivalent algorithmic rate beside it (it may exceed the fp32 peak: that is the point of the kernels)"""
            flops = passes * macs * (S + 2) * B * d
            if not split:
                return mfma(name, [B, d, hid[0]], ms, flops)
            HP = (hid[0] + 15) // 16 * 16
            KP = (HP + 31) // 32 * 32
            chain_passes = passes if passes == 2. else 4.         # backward: recompute + data gradient on bf16, weight gradient fp32
            exe = 6. * chain_passes * (len(hid) - 1) * HP * KP * (S + 2) * B * d
            rows.append({"kernel": name, "shape": [B, d, hid[0]], "ms": round(ms, 4),
                         "bound": "mfma (bf16, 6 terms per fp32 product" + (")" if passes == 2. else "; weight gradient on fp32 MFMA)"),
                         "achieved_TFLOPs_bf16": round(exe / ms / 1e9, 1), "frac_of_2500TF_bf16": round(exe / ms / 1e9 / BF16_PEAK, 3),
                         "fp32_equivalent_TFLOPs": round(flops / ms / 1e9, 1)})
            assert 0. < rows[-1]["frac_of_2500TF_bf16"] <= 1., rows[-1]
        with torch.no_grad():
            mono_row("monotonic_fwd " + tag, time_entry("gnf_monotonic_fwd", lambda: norm(x, h), n=10), 2., split_fwd)
        xg, hg = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
        z, jac = norm(xg, hg)
        gz = torch.randn_like(z)
        ps = list(norm.parameters())
        mono_row("monotonic_bwd " + tag,
                 time_entry("gnf_monotonic_bwd", lambda: torch.autograd.grad((z, jac), [xg, hg] + ps, (gz, gz), retain_graph=True), n=5, warm=1),
                 4., split_bwd)
    # ---- MNISTCNN conv front at cfg4 (78 400 images)
    n = 78400
    e = torch.randn(n, 784, device=DEV).requires_grad_(True)
    W1, b1 = torch.randn(16, 1, 3, 3, device=DEV, requires_grad=True), torch.randn(16, device=DEV, requires_grad=True)
    W2, b2 = torch.randn(16, 16, 3, 3, device=DEV, requires_grad=True), torch.randn(16, device=DEV, requires_grad=True)
    with torch.no_grad():
        mfma("mnistcnn_conv_fwd", [n, 784], time_entry("gnf_mnistcnn_conv_fwd", lambda: ops.MnistConvFn.apply(e, W1, b1, W2, b2), n=10),
             2. * (97344 + 1327104 * 4. / 9.) * n, direct=2. * (97344 + 1327104) * n)
    out = ops.MnistConvFn.apply(e, W1, b1, W2, b2)
    gp = torch.randn_like(out)
    # dW2 + da1 (2 x conv2 MACs) + dW1 + de (2 x conv1 MACs): the same convention as bench.py -- the conv1 RECOMPUTE of the
    # kernel is not algorithmic work (until round 4 this row counted it: 1.025 here against 0.979 in the bench line).  Round 6:
    # the fraction is taken on the EXECUTED basis (dW2 on 144 Winograd tiles, da1 on the 169 tiles of the full correlation)
    mfma("mnistcnn_conv_bwd (dense de: x wants a gradient)", [n, 784],
         time_entry("gnf_mnistcnn_conv_bwd", lambda: torch.autograd.grad(out, (e, W1, b1, W2, b2), gp, retain_graph=True), n=10),
         2. * (1327104 * 4. / 9. * (1. + 169. / 144.) + 2 * 97344) * n, direct=2. * (2 * 1327104 + 2 * 97344) * n)
    for r in rows:
        print(json.dumps(r))
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
