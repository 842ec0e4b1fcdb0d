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



def test_shard_batches_equal_step_counts_for_ragged_sizes():
    """train_uci.shard_batches: every rank runs the same number of steps with the same shard size for any N, b, world
    (a rank with one step fewer would leave its peers waiting in the per-step all-reduce), shards of one global batch
    are disjoint, and together they are the (trimmed) global batch every rank cut identically."""
    import train_uci
    for n, b, world in [(10, 4, 2), (11, 4, 2), (9, 4, 2), (101, 10, 4), (7, 8, 4), (3, 4, 4), (1000, 33, 8)]:
        per_rank = [train_uci.shard_batches(n, b, r, world, torch.Generator().manual_seed(5)) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1, (n, b, world)
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(5))
        for k in range(len(per_rank[0])):
            sizes = {p[k].numel() for p in per_rank}
            assert len(sizes) == 1 and sizes.pop() >= 1
            union = torch.cat([p[k] for p in per_rank])
            assert union.unique().numel() == union.numel()
            chunk = perm[k * b:(k + 1) * b]
            # batches trimmed to nothing are dropped, so index k is the k-th NON-EMPTY global batch
            assert set(union.tolist()) <= set(perm.tolist())
            if n % b == 0 or k < n // b:
                assert set(union.tolist()) == set(chunk[:chunk.numel() // world * world].tolist())


def _ragged_worker(rank, world, port, out):
    import train_uci
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    X = torch.randn(23, 4)                                   # 23 rows, b_size 8, 2 ranks: global batches 8, 8, 7 -> 6
    flow = TinyFlow()
    state = dp.FlatState(flow)
    state.broadcast(0)
    gen = torch.Generator().manual_seed(1234)
    steps = 0
    for rows in train_uci.shard_batches(X.shape[0], 8, rank, world, gen):
        dp.train_step(flow, state, X[rows], lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
        steps += 1
    gathered = [torch.empty_like(state.flat) for _ in range(world)]
    dist.all_gather(gathered, state.flat)                    # reached by every rank: nobody hangs in a step
    if rank == 0:
        out.put((steps, [g.numpy().copy() for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_epoch_with_ragged_dataset_size():
    import train_uci
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    steps, flats = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert steps == 3
    assert (flats[0] == flats[1]).all()
    # equals one process stepping through the same trimmed global batches
    torch.manual_seed(0)
    X = torch.randn(23, 4)
    flow = TinyFlow()
    state = dp.FlatState(flow)
    gen = torch.Generator().manual_seed(1234)
    for rows in train_uci.shard_batches(X.shape[0], 8, 0, 1, gen):
        rows = rows[:rows.numel() // 2 * 2]
        dp.train_step(flow, state, X[rows], lr=1e-2, weight_decay=1e-5, optimizer=torch_adam)
    assert torch.allclose(torch.from_numpy(flats[0]), state.flat, rtol=1e-5, atol=1e-6)


def test_gate_seeds_differ_per_rank_and_per_conditioner():
    seeds = {dp.gate_seed(r, k) for r in range(8) for k in range(4)}
    assert len(seeds) == 32 and all(0 <= s < 2 ** 62 for s in seeds)
    from models import buildFCNormalizingFlow, DAGConditioner, AffineNormalizer
    flow = buildFCNormalizingFlow(3, DAGConditioner, {"in_size": 4, "hidden": [8], "out_size": 2}, AffineNormalizer, {})
    dp.seed_gates(flow, rank=1)
    got = [c.gate_seed for c in flow.getConditioners()]
    assert got == [dp.gate_seed(1, k) for k in range(3)] and len(set(got)) == 3
