"""The three fc1 GEMMs of the headline step (78 400 x 2304 x 128: forward, data gradient, weight gradient) on the fp32-MFMA
kernels (gnf_gemm) and on the split-bf16 kernels (gnf_gemm_split_bf16, classes = 0: the dedicated kernels), alternating on
one box: HIP-event medians, the kernel that ran, max / rms error against an fp64 product.
    python tools/bench_fc1_split.py [fwd|dx|dw ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
if os.environ.get('GNF_AB_LIB'):
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
from gnf_hip import ops
from gnf_hip.abi import ptr, call, stream
dev = "cuda:0"
torch.manual_seed(0)
M0, K0, F0 = 78400, 2304, 128
X = torch.randn(M0, K0, device=dev); W = torch.randn(F0, K0, device=dev) / 48.; dY = torch.randn(M0, F0, device=dev)
b = torch.randn(F0, device=dev)
lib = abi.load()


def f32(A, sa, B, sb, M, N, K, bias=None, relu=False):
    """the fp32-MFMA kernels: gnf_gemm with a workspace of exactly gnf_gemm_f32_ws_bytes -- the split-K partials of the weight
    gradient, nothing for the two single-pass kernels --, which is too small for the split-bf16 dispatch of round 6 (this is
    what GNF_TRUE_F32=1 runs)"""
    C = torch.empty(M, N, device=dev)
    n = int(lib.gnf_gemm_f32_ws_bytes(M, N, K))
    ws = _wsf.setdefault(n, torch.empty(max(n // 4, 1), device=dev))
    call("gnf_gemm", ptr(A), sa[0], sa[1], ptr(B), None, sb[0], sb[1], ptr(C), N, 1, ptr(bias), None, 0, 0, None, 0, 0,
         1 if relu else 0, M, N, K, ptr(ws) if n else None, n, stream())
    return C


_wsf = {}
_ws = {}


def split(A, sa, B, sb, M, N, K, bias=None, relu=False):
    C = torch.empty(M, N, device=dev)
    n = int(lib.gnf_gemm_split_ws_bytes(M, N, K))
    ws = _ws.setdefault(n, torch.empty(max(n, 16), dtype=torch.uint8, device=dev))
    call("gnf_gemm_split_bf16", ptr(A), sa[0], sa[1], ptr(B), sb[0], sb[1], ptr(C), N, 1, ptr(bias), 1 if relu else 0, M, N, K,
         0, 1, 0, abi.rawptr(ws), n, stream())
    return C


def timeit(fn, reps=31):
    for _ in range(60):
        fn()
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); c.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(c))
    ts.sort()
    return ts[len(ts) // 2]


rows = torch.cat([torch.arange(0, 400), torch.randint(0, M0, (3200,)), torch.arange(M0 - 400, M0)]).to(dev)
cases = {
    "fwd": ("fwd  relu(X W^T + b)", (X, (K0, 1), W, (1, K0), M0, F0, K0, b, True),
            lambda C: (C[rows].double(), torch.relu(X[rows].double() @ W.double().t() + b.double()))),
    "dx": ("dX   dY W", (dY, (F0, 1), W, (K0, 1), M0, K0, F0),
           lambda C: (C[rows].double(), dY[rows].double() @ W.double())),
    "dw": ("dW   dY^T X", (dY, (1, F0), X, (K0, 1), F0, K0, M0),
           lambda C: (C.double(), dY.double().t() @ X.double())),
}
fl = 2. * M0 * K0 * F0
for key in (sys.argv[1:] or ["fwd", "dx", "dw"]):
    name, args, chk = cases[key]
    for rnd in range(2):
        for tag, fn, last in (("fp32-MFMA ", f32, lib.gnf_gemm_last_kernel), ("split-bf16", split, lib.gnf_gemm_split_last_kernel)):
            t = timeit(lambda: fn(*args))
            got, ref = chk(fn(*args))
            sc = ref.pow(2).mean().sqrt().item()
            d = got - ref
            print("%-22s %s  %.4f ms  %6.1f TFLOP/s (fp32-equivalent)  kernel %-20s err vs fp64: max %.2e rms %.2e"
                  % (name, tag, t, fl / t / 1e9, last().decode(), d.abs().max().item() / sc, d.pow(2).mean().sqrt().item() / sc), flush=True)
