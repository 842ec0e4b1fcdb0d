"""How long does the HOST take to enqueue one cfg4 training step (Python + ctypes + torch autograd)?"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
import bench
from gnf_hip import dp
flow = bench.build_flow().to("cuda:0")
state = dp.FlatState(flow)
x = bench.pseudo_mnist(torch.Generator().manual_seed(1), 100, 784).to("cuda:0")
for _ in range(3):
    bench.train_step(flow, state, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    bench.train_step(flow, state, x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step: %.2f ms ; wall per step incl. GPU drain: %.2f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
