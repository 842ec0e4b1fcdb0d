"""Micro-driver for rocprofv3: the three fc1 GEMM shapes of cfg4 through gnf_gemm."""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import ops
dev = "cuda:0"
torch.manual_seed(0)
def g(A, B):
    M, K = A.shape; N = B.shape[1]
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, A.stride(), B, B.stride(), C, C.stride(), M, N, K)
    return C
X = torch.randn(78400, 2304, device=dev); W = torch.randn(128, 2304, device=dev); dY = torch.randn(78400, 128, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    g(X, W.t()); g(dY, W); g(dY.t(), X)
torch.cuda.synchronize()
