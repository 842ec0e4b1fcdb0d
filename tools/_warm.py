"""warm_gpu(): ~0.3 s of matrix products before a timing loop.  A GPU that has just been started (or has idled through a
tool's set-up) runs its first tens of milliseconds of kernels at ~2.15 GHz instead of ~2.4 (round 4, DESIGN.md section 6,
"Cold-GPU clocks"): a loop of three warm-up launches and twenty timed ones measures that, not the kernel."""
import time
import torch


def warm_gpu(seconds=.3, dev="cuda:0"):
    a = torch.randn(2048, 2048, device=dev)
    torch.cuda.synchronize()
    t = time.time()
    while time.time() - t < seconds:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()
