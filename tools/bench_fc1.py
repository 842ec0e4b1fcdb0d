"""The three fc1 GEMMs of the headline step (78 400 x 2304 x 128: forward, data gradient, weight gradient) through
gnf_gemm, HIP-event medians.  GNF_AB_LIB=tools/<name>.bin compares another build (tools/build_variant.sh).
    python tools/bench_fc1.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
if os.environ.get('GNF_AB_LIB'):
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
from gnf_hip import ops
dev = "cuda:0"
torch.manual_seed(0)
X = torch.randn(78400, 2304, device=dev); W = torch.randn(128, 2304, device=dev); dY = torch.randn(78400, 128, device=dev)
b = torch.randn(128, device=dev)


def g(A, B, bias=None, relu=False):
    M, K = A.shape; N = B.shape[1]
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, A.stride(), B, B.stride(), C, C.stride(), M, N, K, bias=bias, relu=relu)
    return C


def timeit(fn, reps=41):
    for _ in range(150):          # ~70 ms of load first: a GPU that has just started runs its first kernels at ~2.15 GHz, not 2.4
        fn()
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); c.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(c))
    ts.sort()
    return ts[len(ts) // 2]


fl = 2. * 78400 * 2304 * 128
def err(C, ref):
    return ((C.double() - ref.double()).abs().max() / ref.double().abs().max()).item()


rows = torch.cat([torch.arange(0, 300), torch.randint(0, 78400, (1500,)), torch.arange(78400 - 300, 78400)]).to(dev)
for name, fn, ref in (("fwd  X W^T + b, relu", lambda: g(X, W.t(), b, True), lambda C: err(C[rows], torch.relu(X[rows].double() @ W.double().t() + b.double()))),
                      ("dX   dY W", lambda: g(dY, W), lambda C: err(C[rows], dY[rows].double() @ W.double())),
                      ("dW   dY^T X", lambda: g(dY.t(), X), lambda C: err(C, dY.double().t() @ X.double()))):
    t = timeit(fn)
    print("%-22s %.4f ms  %.1f TFLOP/s   kernel %s   max rel err vs fp64 %.1e"
          % (name, t, fl / t / 1e9, abi.load().gnf_gemm_last_kernel().decode(), ref(fn())))
