"""Compare gnf_gemm with torch.matmul (hipBLASLt/rocBLAS fp32) on the hot GEMM shapes; prints TFLOP/s."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graphical-normalizing-flows_amd")); sys.path.insert(0, ROOT)
from gnf_hip import ops, abi
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.measured_peaks(dev)))

def g(A, B):
    """C = A @ B for arbitrary-stride 2-D views through gnf_gemm."""
    M, K = A.shape; N = B.shape[1]
    C = torch.empty(M, N, device=A.device)
    ops.gemm(A, A.stride(), B, B.stride(), C, C.stride(), M, N, K)
    return C


def timeit(f, n=20):
    for _ in range(150): f()      # warm: the first ~50 ms of kernels of a process run at ~2.15 GHz
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [("fc1 fwd  X[M,K] W[N,K]^T", 78400, 128, 2304, "nt"), ("fc1 dX   dY[M,K] W[K,N]", 78400, 2304, 128, "nn"),
          ("fc1 dW   dY[K,M]^T X[K,N]", 128, 2304, 78400, "tn"), ("square nt", 4096, 4096, 4096, "nt"),
          ("square nn", 4096, 4096, 4096, "nn"), ("MADE cfg5", 50000, 630, 630, "nt"), ("MADE cfg3", 100, 1024, 1024, "nt")]
for name, M, N, K, lay in shapes:
    if lay == "nt":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
        f_t = lambda: A @ B.t()
        f_g = lambda: g(A, B.t())
    elif lay == "nn":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
        f_t = lambda: A @ B
        f_g = lambda: g(A, B)
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
        f_t = lambda: A.t() @ B
        f_g = lambda: g(A.t(), B)
    err = ((f_t() - f_g()).abs().max() / f_t().abs().max()).item()
    tt, tg = timeit(f_t), timeit(f_g)
    fl = 2.0 * M * N * K
    print("%-28s M=%6d N=%5d K=%6d  torch %.3f ms %6.1f TF | gnf %.3f ms %6.1f TF | relerr %.1e"
          % (name, M, N, K, tt, fl / tt / 1e9, tg, fl / tg / 1e9, err), flush=True)
