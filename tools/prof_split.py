"""micro-driver for counter passes over the split-bf16 fc1 kernels:  python tools/prof_split.py fwd|dx|dw [launches]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
from gnf_hip.abi import ptr, call, stream
dev = "cuda:0"
which, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.manual_seed(0)
M0, K0, F0 = 78400, 2304, 128
X = torch.randn(M0, K0, device=dev); W = torch.randn(F0, K0, device=dev) / 48.; dY = torch.randn(M0, F0, device=dev)
lib = abi.load()
args = {"fwd": (X, (K0, 1), W, (1, K0), M0, F0, K0), "dx": (dY, (F0, 1), W, (K0, 1), M0, K0, F0),
        "dw": (dY, (1, F0), X, (K0, 1), F0, K0, M0)}[which]
A, sa, B, sb, M, N, K = args
C = torch.empty(M, N, device=dev)
nws = int(lib.gnf_gemm_split_ws_bytes(M, N, K))
ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
for _ in range(n):
    call("gnf_gemm_split_bf16", ptr(A), sa[0], sa[1], ptr(B), sb[0], sb[1], ptr(C), N, 1, None, 0, M, N, K, 0, 1, 0,
         abi.rawptr(ws), nws, stream())
torch.cuda.synchronize()
print(lib.gnf_gemm_split_last_kernel().decode())
