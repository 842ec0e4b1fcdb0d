"""timing probes of gemm_split_tall_k: real A (HBM stream) vs an A whose rows alias a 10-MB buffer (cache-resident)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
from gnf_hip.abi import ptr, call, stream
dev = "cuda:0"
M0, K0, F0 = 78400, 2304, 128
X = torch.randn(M0, K0, device=dev); W = torch.randn(F0, K0, device=dev) / 48.
lib = abi.load()
C = torch.empty(M0, F0, device=dev)
nws = int(lib.gnf_gemm_split_ws_bytes(M0, F0, K0)); ws = torch.empty(nws, dtype=torch.uint8, device=dev)


def run(sam):
    call("gnf_gemm_split_bf16", ptr(X), sam, 1, ptr(W), 1, K0, ptr(C), F0, 1, None, 0, M0, F0, K0, 0, 1, 0, abi.rawptr(ws), nws, stream())


def timeit(fn, reps=31):
    for _ in range(60): fn()
    ts = []
    for _ in range(reps):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2]


for sam in (K0, 32, K0, 32):
    print("row stride %5d floats: %.4f ms" % (sam, timeit(lambda: run(sam))), lib.gnf_gemm_split_last_kernel().decode())
