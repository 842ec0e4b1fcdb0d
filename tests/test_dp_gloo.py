"""CPU, world_size=2, gloo: the data-parallel step's host logic (flat parameter/gradient
views, ONE all-reduce, replicas stay identical and equal the single-process full-batch
step).  The HIP Adam kernel is replaced by an equivalent torch expression here -- the
kernel itself is covered by the GPU tests; this test is about sharding and the collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from gnf_hip import dp


class TinyFlow(nn.Module):
    """stand-in with the flow's interface: forward -> (z, logdet), loss(z, logdet) with a
    parameter-only constraint term (like the DAG acyclicity loss)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.net = nn.Sequential(nn.Linear(4, 8), nn.Tanh(), nn.Linear(8, 4))
        self.A = nn.Parameter(torch.randn(4, 4) * .1)

    def forward(self, x):
        s = self.net(x)
        return x * torch.exp(s) + (x @ self.A), s.sum(1)

    def loss(self, z, logdet):
        return (self.A ** 2).sum() - (logdet - .5 * (z ** 2).sum(1)).mean()


def torch_adam(state, lr, weight_decay, grad_scale, b1=.9, b2=.999, eps=1e-8):
    g = state.grad * grad_scale + weight_decay * state.flat
    state.m.mul_(b1).add_(g, alpha=1 - b1)
    state.v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** state.t, 1 - b2 ** state.t
    state.flat.sub_(lr / bc1 * state.m / (state.v.sqrt() / bc2 ** .5 + eps))


def _worker(rank, world, port, x, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flow = TinyFlow()
    state = dp.FlatState(flow)
    state.broadcast(0)
    shard = x[rank * (x.shape[0] // world):(rank + 1) * (x.shape[0] // world)]
    for _ in range(3):
        dp.train_step(flow, state, shard, lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
    gathered = [torch.empty_like(state.flat) for _ in range(world)]
    dist.all_gather(gathered, state.flat)
    if rank == 0:
        out.put([g.numpy().copy() for g in gathered])      # by value: a shared-memory tensor handle can die with the worker
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process():
    torch.manual_seed(0)
    x = torch.randn(16, 4)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, q)) for r in range(2)]
    for p in procs:
        p.start()
    flats = [torch.from_numpy(a) for a in q.get()]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(flats[0], flats[1])                 # replicas identical after 3 steps
    flow = TinyFlow()
    state = dp.FlatState(flow)
    for _ in range(3):
        dp.train_step(flow, state, x, lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
    assert torch.allclose(flats[0], state.flat, rtol=1e-5, atol=1e-6)   # == full-batch single process
    for p in flow.parameters():                           # parameters really are views of the flat buffer
        assert p.data.data_ptr() >= state.flat.data_ptr()
        assert p.data.data_ptr() < state.flat.data_ptr() + state.flat.numel() * 4


def test_adam_checkpoint_round_trips_through_torch_optim():
    """FlatState.optimizer_state_dict() is a valid torch.optim.Adam state: a torch Adam resumed from it takes the same
    next step as the flat optimiser, and loading it back restores the moments (ADAM.pt compatibility with the
    reference drivers, UCIExperiments.py:216-220)."""
    torch.manual_seed(1)
    x = torch.randn(32, 4)
    flow = TinyFlow()
    state = dp.FlatState(flow)
    for _ in range(3):
        dp.train_step(flow, state, x, lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
    sd = state.optimizer_state_dict(flow, lr=1e-2, weight_decay=1e-5)
    # a stock torch model + Adam resumed from the exported files
    ref = TinyFlow()
    ref.load_state_dict({k: v.clone() for k, v in flow.state_dict().items()})
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-5)
    opt.load_state_dict(sd)
    z, ld = ref(x)
    opt.zero_grad()
    ref.loss(z, ld).backward()
    opt.step()
    dp.train_step(flow, state, x, lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
    for (k, a), (_, b) in zip(flow.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), k
    # and back: a fresh FlatState picks the moments and the step count up again
    flow2 = TinyFlow()
    flow2.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    state2 = dp.FlatState(flow2)
    state2.load_optimizer_state_dict(flow2, opt.state_dict())
    assert state2.t == 4
    assert torch.allclose(state2.m, state.m, rtol=1e-5, atol=1e-8) and torch.allclose(state2.v, state.v, rtol=1e-5, atol=1e-10)

