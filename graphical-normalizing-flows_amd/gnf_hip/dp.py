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

    def pack_grads(self):
        """flat gradient buffer <- the .grad tensors autograd just produced, in ONE concatenation launch.  (.grad views
        into the flat buffer would cost a zero-fill plus one accumulate kernel per parameter per step.)"""
        parts = []
        for p, pad in zip(self.params, self._pads):
            parts.append(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device=pad.device))
            if pad.numel():
                parts.append(pad)
        torch.cat(parts, out=self.grad)
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
