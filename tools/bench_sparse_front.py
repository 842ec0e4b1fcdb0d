"""Deterministic-gate embedding front at the cfg4 size (B = 100 -> 78 400 masked copies): the sparse crop path
(gnf_mnistcnn_sparse_fwd / _bwd) against the dense kernels (gate + Winograd conv + fc1): conditioner forward under
no_grad, and forward + backward w.r.t. the network parameters (frozen binary A, as after post_process()).
Usage: python tools/bench_sparse_front.py [B]"""
import json
import sys

import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi  # noqa: E402
from models import DAGConditioner  # noqa: E402
from models.MLP import MNISTCNN  # noqa: E402
from models.NormalizingFlowFactories import MNIST_A_prior  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.manual_seed(0)
    cond = DAGConditioner(784, MNISTCNN(out_d=30), 30, A_prior=MNIST_A_prior(28, 2)).to(DEV)
    cond.stoch_gate = False
    x = torch.rand(B, 784, device=DEV)
    out = {"B": B, "masked_copies": B * 784}
    with torch.no_grad():
        hs = cond(x)
        cond.sparse_front = False
        hd = cond(x)
        out["max_abs_diff"] = float((hs - hd).abs().max())
        out["dense_ms"] = timed(lambda: cond(x))
        cond.sparse_front = True
        out["sparse_ms"] = timed(lambda: cond(x))
        abi.profile_enable(["gnf_mnistcnn_sparse_fwd", "gnf_gemm"])
        cond(x)
        out["sparse_entry_ms"] = {k: round(v, 4) for k, v in abi.profile_collect().items()}
    out["speedup"] = out["dense_ms"] / out["sparse_ms"]
    # training with the frozen gate
    cond.s_thresh = False
    cond.A.requires_grad = False
    gh = torch.randn(B, 784, 30, device=DEV)

    def step():
        for p in cond.parameters():
            p.grad = None
        (cond(x) * gh).sum().backward()
    out["sparse_fwd_bwd_ms"] = timed(step)
    abi.profile_enable(["gnf_mnistcnn_sparse_fwd", "gnf_mnistcnn_sparse_bwd", "gnf_mnistcnn_sparse_bwd_tables"])
    step()
    out["sparse_train_entry_ms"] = {k: round(v, 4) for k, v in abi.profile_collect().items()}
    cond.sparse_front = False
    out["dense_fwd_bwd_ms"] = timed(step)
    out["train_speedup"] = out["dense_fwd_bwd_ms"] / out["sparse_fwd_bwd_ms"]
    # algorithmic work of the sparse path: conv1 12*12*16*9 + conv2 10*10*16*16*9 + fc1 400*128 MAC per copy
    mac = 12 * 12 * 16 * 9 + 10 * 10 * 16 * 16 * 9 + 400 * 128
    out["sparse_TFLOPs"] = 2 * mac * B * 784 / (out["sparse_ms"] * 1e-3) / 1e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
