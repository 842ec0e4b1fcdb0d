"""Batch data-parallel training step, one process per GPU (replaces the reference's
single-process nn.DataParallel, ImageExperiments.py:168,199-216).

Every rank holds an identical replica and a shard of the minibatch.  All trainable
parameters and their gradients are views of two flat fp32 buffers, so a step needs exactly
ONE collective -- an all-reduce(sum) of the flat gradient buffer over RCCL/xGMI -- followed
by one fused Adam launch.  The DAG acyclicity term depends on parameters only: every rank
computes it redundantly, so after averaging it is counted once, like the reference where it is
added once on GPU 0."""
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
        self._pads = []                            # zero fillers between parameters, for the one-launch gradient pack
        for p in self.params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p)
            p.grad = None
            self._pads.append(torch.zeros(pad4(k) - k, device=dev))
            o += pad4(k)
        self.t = 0

    def pack_grads(self, grads=None):
        """flat gradient buffer <- the .grad tensors autograd just produced (or the given list, in self.params order),
        in ONE concatenation launch.  (.grad views into the flat buffer would cost a zero-fill plus one accumulate
        kernel per parameter per step.)"""
        parts = []
        for i, (p, pad) in enumerate(zip(self.params, self._pads)):
            g = p.grad if grads is None else grads[i]
            parts.append(g.reshape(-1) if g is not None else torch.zeros(p.numel(), device=pad.device))
            if pad.numel():
                parts.append(pad)
        torch.cat(parts, out=self.grad)
        if grads is None:
            for p in self.params:
                p.grad = None

    def broadcast(self, src=0):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat, src)

    # ---- checkpoint compatibility with the reference drivers, which save `torch.optim.Adam(model.parameters())
    #      .state_dict()` as ADAM.pt (UCIExperiments.py:216-220, ImageExperiments.py:251-253)
    def optimizer_state_dict(self, module, lr=1e-3, weight_decay=1e-5):
        index = {id(p): i for i, p in enumerate(module.parameters())}
        state, o = {}, 0
        for p in self.params:
            k = p.numel()
            if self.t > 0:
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
            st = sd["state"].get(index[id(p)])
            if st is not None:
                self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.t = max(self.t, int(float(st["step"])))
            o += (k + 3) // 4 * 4


def hip_adam(state, lr, weight_decay, grad_scale):
    from . import ops
    ops.adam_step(state.flat, state.grad, state.m, state.v, state.t, lr=lr, weight_decay=weight_decay,
                  grad_scale=grad_scale)


def train_step(flow, state, x_shard, lr=1e-3, weight_decay=1e-5, optimizer=hip_adam):
    """fwd + log|det J| + NLL (+ constraints) + bwd on the local shard, one all-reduce, Adam.
    loss_rank = constraints - mean_local(log p); averaging over ranks gives the global mean."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    z, logdet = flow(x_shard)
    loss = flow.loss(z, logdet)
    loss.backward()
    state.pack_grads()
    if world > 1:
        dist.all_reduce(state.grad)                 # the step's only collective
    state.t += 1
    optimizer(state, lr, weight_decay, 1. / world)
    return loss


class GraphedStep:
    """The whole optimisation step (fwd + log|det J| + NLL + bwd + gradient pack + Adam) captured ONCE into a hipGraph
    and replayed: for the launch-bound configurations (toy d=2, MADE at B=100: ~100 launches of a few microseconds,
    host enqueue ~0.8 ms per step) the replay removes the per-launch host cost.  Single-GPU only (the all-reduce stays
    eager); nothing step-dependent may be passed to a kernel by value, so Adam's step count lives in device memory
    (gnf_adam_step_dev).  Flows whose conditioner draws Philox noise from a host-side call counter (stochastic DAG
    gate) would replay the same noise: they are refused.

    Construction runs `warmup` real steps on x_example (they count as training steps), then captures."""

    def __init__(self, flow, state, x_example, lr=1e-3, weight_decay=1e-5, warmup=3):
        from . import ops
        if dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedStep is single-process; use train_step under torchrun")
        for c in flow.getConditioners():
            if getattr(c, "stoch_gate", False) or getattr(c, "noise_gate", False):
                if getattr(c, "h_thresh", 0) > 0 or getattr(c, "s_thresh", False):
                    raise RuntimeError("a stochastic DAG gate draws its noise from a host-side counter: not graphable")
        self.state, self.flow = state, flow
        self.x = x_example.clone()
        self.step_dev = torch.full((1,), state.t, dtype=torch.int32, device=self.x.device)

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

        def body():
            leaves = [p.detach().requires_grad_() for p in state.params]
            loss = torch.func.functional_call(wrapper, dict(zip(names, leaves)), (self.x,))
            state.pack_grads(torch.autograd.grad(loss, leaves, allow_unused=True))
            ops.adam_step_dev(state.flat, state.grad, state.m, state.v, self.step_dev, lr=lr,
                              weight_decay=weight_decay)
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                body()
                state.t += 1
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = body()

    def __call__(self, x):
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        self.state.t += 1
        return self.loss

