"""Review of round 5, item 7: the Affine normalizer at the BASELINE shape [50 000, 63] (cfg5's row count and width) against the
floors of its launch shape -- an EMPTY grid of the same shape, and the STREAM copy kernel moving the same number of bytes --
all timed the same way (HIP events around ONE launch on an idle stream, median of 41; and back to back, 50 launches per event pair).
    python tools/bench_affine_floor.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops
from _warm import warm_gpu
dev = "cuda:0"
lib = abi.load()
st = abi.stream


def single(fn, reps=41):
    warm_gpu(.15)
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2] * 1e3


def b2b(fn, n=50, reps=9):
    warm_gpu(.15)
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c) / n)
    ts.sort(); return ts[len(ts) // 2] * 1e3


for B, d in [(50000, 63), (1000000, 63)]:
    x = torch.randn(B, d, device=dev); h = torch.randn(B, d, 2, device=dev)
    z, ld = torch.empty(B, d, device=dev), torch.empty(B, device=dev)
    gz, gl = torch.randn(B, d, device=dev), torch.randn(B, device=dev)
    gx, gh = torch.empty(B, d, device=dev), torch.empty(B, d, 2, device=dev)
    fb, bb = 16. * B * d + 4 * B, 28. * B * d + 4 * B
    U = 4 if B * d >= (1 << 24) else 1                       # gnf_affine_fwd / _bwd's rule (gnf_rowwise.hip)
    grid = min((B + 16 * U - 1) // (16 * U), 4096)
    P = abi.ptr
    # the C-ABI entry points themselves (preallocated outputs: no autograd bookkeeping, no allocator)
    def fwd(): abi.call("gnf_affine_fwd", P(x), P(h), 2 * d, 2, 1, P(z), None, P(ld), None, 0, B, d, st())
    def bwd(): abi.call("gnf_affine_bwd", P(x), P(h), 2 * d, 2, 1, P(gz), None, P(gl), None, P(gx), P(gh), 2 * d, 2, 1, B, d, st())
    src = torch.empty(int(bb) // 8 // 4 * 4 + 4, device=dev); dst = torch.empty_like(src)
    nf, nb = int(fb) // 8 // 4 * 4, int(bb) // 8 // 4 * 4
    rows = [("gnf_affine_fwd (z, logdet)  U=%d" % U, fwd, fb), ("gnf_affine_bwd  U=%d" % U, bwd, bb),
            ("empty grid %d x 256" % grid, lambda: lib.gnf_probe_empty(grid, 256, st()), 0.),
            ("probe_copy, bytes of the forward", lambda: lib.gnf_probe_copy(abi.ptr(dst), abi.ptr(src), nf, st()), 8. * nf),
            ("probe_copy, bytes of the backward", lambda: lib.gnf_probe_copy(abi.ptr(dst), abi.ptr(src), nb, st()), 8. * nb)]
    for name, fn, nbytes in rows:
        t1, t2 = single(fn), b2b(fn)
        print("[%7d x %d] %-36s single launch %6.2f us   back to back %6.2f us   %s" % (B, d, name, t1, t2,
              ("%.3f / %.3f of 8 TB/s" % (nbytes / t1 / 1e6 / 8., nbytes / t2 / 1e6 / 8.)) if nbytes else ""), flush=True)
