"""Batch data-parallel training step, one process per GPU (replaces the reference's
single-process nn.DataParallel, ImageExperiments.py:168,199-216).

Every rank holds an identical replica and a shard of the minibatch.  All trainable
parameters and their gradients are views of two flat fp32 buffers, so a step needs exactly
ONE collective -- an all-reduce(sum) of the flat gradient buffer over RCCL/xGMI -- followed
by one fused Adam launch.  The DAG acyclicity term depends on parameters only: every rank
computes it redundantly, so after averaging it is counted once, like the reference where it is
added once on GPU 0."""
import gc
import os
import weakref

import torch
import torch.distributed as dist


class FlatState:
    """Parameters / gradients / Adam moments of `module` as views of flat buffers."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        pad4 = lambda k: (k + 3) // 4 * 4          # every parameter starts 16-B aligned (vector loads)
        n = sum(pad4(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        o = 0
        for p in self.params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p)
            p.grad = None
            o += pad4(k)
        self.t = 0
        self._offsets = []
        o = 0
        for q in self.params:
            self._offsets.append(o)
            o += pad4(q.numel())
        self.active_runs = [(0, n)]                # [(offset, length)] of the flat buffer Adam steps; see pack_grads
        # the slot of every parameter's gradient in the flat buffer: backward kernels write there directly (ops.grad_out)
        self.grad_views = [self.grad[o:o + p.numel()] for o, p in zip(self._offsets, self.params)]
        self._taken = set()
        from . import ops
        ops.register_grad_slots(self, self.params)

    def __del__(self):
        try:
            from . import ops
            ops.unregister_grad_slots(self)
        except Exception:                          # noqa: BLE001  (interpreter shutdown)
            pass

    def pack_grads(self, grads=None):
        """flat gradient buffer <- the .grad tensors autograd just produced (or the given list, in self.params order).
        Gradients the backward kernels already wrote into their slot (ops.grad_out) stay where they are; the others are
        copied in one multi-tensor launch.  (.grad views into the flat buffer for everything would cost a zero-fill plus
        one accumulate kernel per parameter per step.)"""
        dst, src, runs, absent = [], [], [], 0
        for i, (p, slot) in enumerate(zip(self.params, self.grad_views)):
            g = p.grad if grads is None else grads[i]
            if g is None:
                slot.zero_()                       # e.g. a frozen A: the all-reduce must not sum stale values
                absent += 1
                continue
            if not (g.data_ptr() == slot.data_ptr() and g.is_contiguous()):
                dst.append(slot)
                src.append(g.reshape(-1))
            # torch.optim.Adam skips parameters without a gradient (no weight decay, no moment update): a frozen A
            # (post_process) must not decay.  Adam therefore steps the runs of consecutive parameters that have one.
            o, k = self._offsets[i], (p.numel() + 3) // 4 * 4
            if runs and runs[-1][0] + runs[-1][1] == o:
                runs[-1] = (runs[-1][0], runs[-1][1] + k)
            else:
                runs.append((o, k))
        self.active_runs = runs
        self.pack_stats = {"in_place": len(self.params) - len(dst) - absent, "copied": len(dst), "absent": absent}
        if dst:
            torch._foreach_copy_(dst, src)
        self._taken.clear()
        if grads is None:
            for p in self.params:
                p.grad = None

    def drop_grads(self):
        """forget every pending gradient (the reference's zero_grad() between accumulation windows): clears .grad AND the
        record of which flat-buffer slots the backward kernels handed out, so the next backward writes in place again"""
        for p in self.params:
            p.grad = None
        self._taken.clear()

    def broadcast(self, src=0):
        if collective_on():
            if dist.get_backend() == "gloo" and self.flat.is_cuda:
                h = self.flat.cpu()
                dist.broadcast(h, src)
                self.flat.copy_(h)
            else:
                dist.broadcast(self.flat, src)

    # ---- checkpoint compatibility with the reference drivers, which save `torch.optim.Adam(model.parameters())
    #      .state_dict()` as ADAM.pt (UCIExperiments.py:216-220, ImageExperiments.py:251-253)
    def optimizer_state_dict(self, module, lr=1e-3, weight_decay=1e-5):
        index = {id(p): i for i, p in enumerate(module.parameters())}
        state, o = {}, 0
        for p in self.params:
            k = p.numel()
            if self.t > 0 and id(p) in index:
                state[index[id(p)]] = {"step": torch.tensor(float(self.t)),
                                       "exp_avg": self.m[o:o + k].view_as(p).clone(),
                                       "exp_avg_sq": self.v[o:o + k].view_as(p).clone()}
            o += (k + 3) // 4 * 4
        group = {"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(index)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, module, sd):
        index = {id(p): i for i, p in enumerate(module.parameters())}
        o = 0
        for p in self.params:
            k = p.numel()
            st = sd["state"].get(index.get(id(p)))
            if st is not None:
                self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.t = max(self.t, int(float(st["step"])))
            o += (k + 3) // 4 * 4


def hip_adam(state, lr, weight_decay, grad_scale):
    from . import ops
    for o, k in state.active_runs:                  # one launch unless some parameter is frozen mid-buffer
        ops.adam_step(state.flat[o:o + k], state.grad[o:o + k], state.m[o:o + k], state.v[o:o + k], state.t, lr=lr,
                      weight_decay=weight_decay, grad_scale=grad_scale)


def gate_seed(rank, k, base=0x9E3779B97F4A7C15):
    """Philox key of conditioner k on rank `rank` (splitmix64 of the pair): replicas draw different gate noise per
    shard like nn.DataParallel's replicas, and the steps of one flow draw independent noise like the reference's
    per-call torch.rand (DAGConditioner.py:99-100)."""
    m = (1 << 64) - 1
    z = (base + 0xBF58476D1CE4E5B9 * (rank + 1) + 0x94D049BB133111EB * (k + 1)) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return (z ^ (z >> 31)) & ((1 << 62) - 1)


def seed_gates(flow, rank):
    for k, c in enumerate(flow.getConditioners()):
        if hasattr(c, "gate_seed"):
            c.gate_seed = gate_seed(rank, k)


def collective_on():
    """the gradient all-reduce runs when there is more than one rank -- or when GNF_FORCE_DIST=1 asks for it at world
    size 1, so that the RCCL branch (process group on the device, in-place device all-reduce, device all-gather of the
    replica checksums, barrier) can be executed and timed on a box with ONE GPU"""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("GNF_FORCE_DIST") == "1"


comm_events = None      # [(start, end)] HIP events around the step's all-reduce while bench.py has profiling on


def comm_profile(enable):
    """start / stop recording the all-reduce time; stop returns the mean milliseconds per call (None without calls)"""
    global comm_events
    if enable:
        comm_events = []
        return None
    evs, comm_events = comm_events, None
    if not evs:
        return None
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / len(evs)


def all_reduce_sum(t):
    """the step's collective: RCCL all-reduce in place; under the gloo backend (CPU transport, used only to exercise
    the N>1 path on boxes with one GPU) the buffer is staged through host memory."""
    ev = None
    if comm_events is not None and t.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend() == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    if ev is not None:
        ev[1].record()
        comm_events.append(ev)


def replicas_identical(state, module=None):
    """True when every rank holds bit-identical parameters (and, given the module, buffers -- the DAG dual variables
    live there): an order-independent integer checksum of the raw fp32 bits, all-gathered."""
    bits = state.flat.view(torch.int32).to(torch.int64).sum()
    if module is not None:
        for b in module.buffers():
            if b.dtype == torch.float32 and b.numel():
                bits = bits + b.detach().contiguous().view(torch.int32).to(torch.int64).sum().to(bits.device) * 31
    if not collective_on():
        return True
    mine = bits.reshape(1)
    if dist.get_backend() == "gloo":
        mine = mine.cpu()
    all_ = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(all_, mine)
    return all(int(a.item()) == int(all_[0].item()) for a in all_)


GRAPH_MAX_ELEMS = 1 << 18               # auto-replay only steps small enough to be launch-bound (cfg1: 1 Ki, cfg3: 77 Ki
                                        # elements; cfg5's 3.1 M-element step is GPU-bound and stages GiBs per capture)


def train_step(flow, state, x_shard, lr=1e-3, weight_decay=1e-5, optimizer=hip_adam, graph="auto"):
    """fwd + log|det J| + NLL (+ constraints) + bwd on the local shard, one all-reduce, Adam.
    loss_rank = constraints - mean_local(log p); averaging over ranks gives the global mean.

    graph="auto": single-process steps of gate-free (graphable) flows on small batches are captured once per variant and
    replayed as a hipGraph (GraphedStep: cfg1 0.68 -> 0.16 ms, cfg3 0.72 -> 0.27 ms per step); graph=False forces the
    launch-by-launch path.  The returned loss tensor of a replayed step is reused by the next replay."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if (graph and world == 1 and not collective_on() and optimizer is hip_adam and x_shard.is_cuda
            and x_shard.numel() <= GRAPH_MAX_ELEMS and torch.is_grad_enabled() and GraphedStep.graphable(flow)):
        gs = getattr(state, "_graphed", None)
        if gs is None or gs.flow is not flow or (gs.lr, gs.weight_decay) != (lr, weight_decay):
            state._graphed = gs = GraphedStep(flow, state, x_shard, lr=lr, weight_decay=weight_decay, warmup=1,
                                              owned_by_state=True)
            return gs.loss
        return gs(x_shard)
    loss = accumulate(flow, x_shard)
    apply_step(state, lr, weight_decay, optimizer)
    return loss


_ONES = {}


def _one(like):
    """the cotangent of the scalar loss, kept per device: autograd would fill a fresh one per step (one launch)"""
    key = (like.device, like.dtype)
    if key not in _ONES:
        _ONES[key] = torch.ones((), device=like.device, dtype=like.dtype)
    return _ONES[key]


def accumulate(flow, x_shard, scale=1.):
    """fwd + log|det J| + NLL (+ constraints) + bwd of ONE micro-batch; gradients add up in `.grad` until apply_step().
    The image driver's gradient accumulation (ImageExperiments.py:205-213: loss / batch_per_optim_step, backward on
    every batch, optimiser every k-th) is k calls with scale = 1/k followed by one apply_step."""
    z, logdet = flow(x_shard)
    loss = flow.loss(z, logdet)
    if scale != 1.:
        loss = loss * scale
    loss.backward(_one(loss) if loss.dim() == 0 else None)
    return loss


def apply_step(state, lr=1e-3, weight_decay=1e-5, optimizer=hip_adam):
    """the accumulated gradients into the flat buffer, ONE all-reduce, Adam"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    state.pack_grads()
    if collective_on():
        all_reduce_sum(state.grad)                  # the step's only collective
    state.t += 1
    optimizer(state, lr, weight_decay, 1. / world)


class GraphedStep:
    """The whole optimisation step (fwd + log|det J| + NLL + bwd + gradient pack + Adam) captured into a hipGraph and
    replayed: for the launch-bound configurations (toy d=2, MADE at B=100: ~100 launches of a few microseconds, host
    enqueue ~0.8 ms per step) the replay removes the per-launch host cost.  Single-GPU only (the all-reduce stays
    eager); nothing step-dependent may be passed to a kernel by value, so Adam's step count lives in device memory
    (gnf_adam_step_dev).  Flows whose conditioner draws Philox noise from a host-side call counter (stochastic DAG
    gate) would replay the same noise: they are refused.

    Host-side state that a captured step bakes in BY VALUE -- a DAG conditioner's matrix-power exponent and gate
    flags, which `model.step()` / `update_dual_param()` / `post_process()` change at epoch level, whether A is
    trainable, a Monotonic normalizer's node count, the batch shape -- is fingerprinted; when the fingerprint differs
    from the captured one the step is re-captured (the dual variables themselves are device buffers updated in place,
    so a replay reads their current values).

    Construction runs `warmup` real steps on x_example (they count as training steps), then captures."""

    MAX_GRAPHS = 16                      # captured variants kept (node-count jitter 20..29 of the drivers = 10)

    def __init__(self, flow, state, x_example, lr=1e-3, weight_decay=1e-5, warmup=3, owned_by_state=False):
        if dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedStep is single-process; use train_step under torchrun")
        # owned_by_state (train_step keeps this object in `state._graphed`): only a weak reference back, so that the pair
        # is NOT a reference cycle -- a cycle is freed by the garbage collector at an arbitrary later moment, and destroying
        # captured graphs while ANOTHER capture is in progress aborts the process (hipGraphDestroy inside a capturing
        # stream).  A GraphedStep the caller builds itself keeps its state alive.
        self._state_ref = weakref.ref(state)
        self._state_keep = None if owned_by_state else state
        self.flow = flow
        self.lr, self.weight_decay = lr, weight_decay
        self._step_buf = torch.zeros(2, dtype=torch.int32, device=x_example.device)    # {steps taken, ticket counter}
        self.step_dev = self._step_buf[:1]
        self.captures = 0
        self._graphs = {}                    # fingerprint -> (graph, loss tensor, input buffer)
        self.loss = self._capture(x_example, max(warmup, 1))

    @property
    def state(self):
        st = self._state_ref()
        if st is None:
            raise RuntimeError("the FlatState of this GraphedStep is gone")
        return st

    @staticmethod
    def graphable(flow):
        for c in flow.getConditioners():
            if getattr(c, "stoch_gate", False) or getattr(c, "noise_gate", False):
                if getattr(c, "h_thresh", 0) > 0 or getattr(c, "s_thresh", False):
                    return False
        return True

    def _fingerprint(self, shape):
        fp = [tuple(shape)]
        for c in self.flow.getConditioners():
            fp.append((getattr(c, "exponent", None), getattr(c, "stoch_gate", None), getattr(c, "noise_gate", None),
                       getattr(c, "s_thresh", None), float(getattr(c, "h_thresh", 0.)), getattr(c, "alpha_factor", None),
                       tuple((id(b), b._version) for b in c.buffers(recurse=False)),    # (version: the constraint term of a
                       getattr(c, "_cache_epoch", 0)))
            # frozen gate is a constant baked into the capture, DAGConditioner.loss)
        for nrm in self.flow.getNormalizers():
            fp.append(getattr(nrm, "nb_steps", None))
        fp.append(tuple(p.requires_grad for p in self.state.params))
        return tuple(fp)

    def _capture(self, x, warmup):
        """`warmup` eager steps on x (real, counted training steps; the loss of the last one is returned), then the
        capture of one more step for this fingerprint"""
        from . import ops
        flow, state = self.flow, self.state
        if not self.graphable(flow):
            raise RuntimeError("a stochastic DAG gate draws its noise from a host-side counter: not graphable")
        lr, weight_decay = self.lr, self.weight_decay
        xbuf = x.clone()

        # The captured step differentiates w.r.t. FRESH leaves aliasing the parameters (torch.func.functional_call), not
        # the nn.Parameters themselves: a Parameter's gradient accumulator remembers the stream it was created on and is
        # kept alive by any older graph the caller still references (a stored loss); capturing through it would record a
        # dependency on the default stream and the HIP runtime crashes in hipGraphInstantiate.
        class _Loss(torch.nn.Module):
            def __init__(self, f):
                super().__init__()
                self.flow = f

            def forward(self, x):
                z, logdet = self.flow(x)
                return self.flow.loss(z, logdet)

        wrapper = _Loss(flow)
        name_of = {id(p): n for n, p in wrapper.named_parameters()}
        names = [name_of[id(p)] for p in state.params]
        live = [i for i, p in enumerate(state.params) if p.requires_grad]     # a frozen A stays out of the backward

        def body():
            leaves = [p.detach().requires_grad_(i in live) for i, p in enumerate(state.params)]
            loss = torch.func.functional_call(wrapper, dict(zip(names, leaves)), (xbuf,))
            got = torch.autograd.grad(loss, [leaves[i] for i in live], grad_outputs=one, allow_unused=True)
            grads = [None] * len(leaves)
            for i, g in zip(live, got):
                grads[i] = g
            state.pack_grads(grads)
            runs = state.active_runs
            for k, (o, n) in enumerate(runs):       # the device-side counter is advanced by the last launch only
                ops.adam_step_dev(state.flat[o:o + n], state.grad[o:o + n], state.m[o:o + n], state.v[o:o + n],
                                  self._step_buf, lr=lr, weight_decay=weight_decay, advance=(k == len(runs) - 1))
            return loss.detach()

        one = _one(xbuf)
        self.step_dev.fill_(state.t)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                last = body().clone()
                state.t += 1
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()                            # no finaliser (of some older graph, event, ...) may run inside the capture
        try:
            with torch.cuda.graph(graph):
                loss = body()
        finally:
            if gc_was_on:
                gc.enable()
        if len(self._graphs) >= self.MAX_GRAPHS:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[self._fingerprint(x.shape)] = (graph, loss, xbuf)
        self.captures += 1
        self._dev_t = state.t                   # host mirror of step_dev (a capture itself executes nothing)
        return last

    def __call__(self, x):
        entry = self._graphs.get(self._fingerprint(x.shape))
        if entry is None:
            # something a captured step bakes in by value changed (or a new batch shape / node count): this call's step
            # runs eagerly and a graph for the new variant is captured behind it
            self.loss = self._capture(x, 1)
            return self.loss
        graph, loss, xbuf = entry
        xbuf.copy_(x, non_blocking=True)
        if getattr(self, "_dev_t", None) != self.state.t:
            # eager steps in between (graph=False, a batch above GRAPH_MAX_ELEMS, a variant that was briefly not
            # graphable) advanced state.t but not the device-side counter the replayed Adam launches read: re-sync it
            # with a plain launch outside the graph, or the bias corrections lr/(1-b1^t), sqrt(1-b2^t) go stale
            self.step_dev.fill_(self.state.t)
        graph.replay()
        self.state.t += 1
        self._dev_t = self.state.t
        self.loss = loss
        return loss
