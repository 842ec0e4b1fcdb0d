import networkx as nx
import numpy as np
import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditioner import Conditioner, linear_pairs, no_context


class DAGMLP(nn.Module):
    """Parameter container, same layout/keys as reference DAGConditioner.py:7-20."""

    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super(DAGMLP, self).__init__()
        l1 = [in_size + cond_in] + hidden
        l2 = hidden + [out_size]
        layers = []
        for h1, h2 in zip(l1, l2):
            layers += [nn.Linear(h1, h2), nn.ReLU()]
        layers.pop()
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return ops.mlp(x, linear_pairs(self.net))


def _digraph(adj):
    """networkx >= 3 spelling of the reference's nx.from_numpy_matrix(..., create_using=nx.DiGraph)."""
    return nx.from_numpy_array(adj, create_using=nx.DiGraph)


class DAGConditioner(Conditioner):
    """Graphical conditioner (reference DAGConditioner.py:23-293): a learned adjacency A gates
    which inputs each dimension's embedding may see; an acyclicity constraint with dual
    variables is added to the loss.

    Forward = fused gate kernel (soft/hard threshold of A, Gumbel-softmax or noise gate with
    on-device Philox noise, masked expand, optional one-hot) -> embedding net.  The
    epoch-level control logic (step / update_dual_param / post_process) is host Python as in
    the reference.  `gate_noise = (u1, u2)` injects explicit uniforms for parity tests."""

    def __init__(self, in_size, hidden, out_size, cond_in=0, soft_thresholding=True, h_thresh=0., gumble_T=1.,
                 hot_encoding=False, l1=0., nb_epoch_update=1, A_prior=None):
        super(DAGConditioner, self).__init__()
        if A_prior is None:
            self.A = nn.Parameter(torch.ones(in_size, in_size) * 1.5 + torch.randn((in_size, in_size)) * .02)
        else:
            self.A = nn.Parameter(A_prior)
        self.in_size = in_size
        self.cond_in = cond_in
        self.exponent = self.in_size % 50
        self.s_thresh = soft_thresholding
        self.h_thresh = h_thresh
        self.stoch_gate = True
        self.noise_gate = False
        in_net = in_size * 2 if hot_encoding else in_size
        if issubclass(type(hidden), nn.Module):
            self.embedding_net = hidden
        else:
            self.embedding_net = DAGMLP(in_net, hidden, out_size, cond_in)
        self.gumble = True
        self.hutchinson = False
        self.gumble_T = gumble_T
        self.hot_encoding = hot_encoding
        with torch.no_grad():
            self.constrainA(h_thresh)
        # dual variables of the acyclicity constraint (same buffer names as the reference)
        self.register_buffer("lambd", torch.tensor(.0))
        self.register_buffer("c", torch.tensor(1e-3))
        self.register_buffer("eta", torch.tensor(10.))
        self.register_buffer("gamma", torch.tensor(.9))
        self.register_buffer("l1_weight", torch.tensor(l1))
        self.register_buffer("dag_const", torch.tensor(1.))
        self.alpha_factor = 1.
        self.d = in_size
        self.tol = 1e-30
        self.register_buffer("alpha", self.getAlpha())
        self.register_buffer("prev_trace", self.get_power_trace())
        self.nb_epoch_update = nb_epoch_update
        self.no_update = 0
        self.is_invertible = False
        self.sparse_front = True        # deterministic gate + windowed, frozen A: sparse embedding front
        self.fused_front = True         # gate + MNISTCNN conv front as one autograd node (False: DagGateFn + MnistConvFn)
        self._sparse_outside = None     # 1 outside the 5x5 pixel windows (device mask, built on first use)
        self._sparse_checked = (None, False)
        self._off_key, self._off = None, False
        self._frozen_loss = (None, None)   # (state key, value) of the constraint term while A is frozen
        self._cache_epoch = 0              # bumped by invalidate_caches(): part of every value-dependent cache key
        self._sparse_plans = {}
        self.gate_noise = None          # (u1, u2) [B,d,d] uniforms (test hook); None -> Philox
        self._gate_calls = 0
        self.gate_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

    def invalidate_caches(self):
        """Forget everything this conditioner remembers ABOUT THE VALUES of A and of its dual buffers: the "A is windowed"
        verdict of the sparse front, the "constraints are switched off" verdict, the constraint term of a frozen gate,
        and (through `_cache_epoch`, part of their keys) the level schedule of NormalizingFlowStep.invert, its captured
        sampling graphs and the captured steps of gnf_hip.dp.GraphedStep.

        These caches are keyed on autograd's version counters and on storage addresses, which every in-place torch op
        moves -- but a write through `.data` (`cond.A.data.mul_(.5)`, the idiom of the reference's own
        DAGConditioner.py:89 and of torch-1.5-era callers) or through a raw pointer moves neither.  Every writer inside
        this package calls this method (post_process, constrainA, the dual update, load_state_dict, the epoch-level
        step()); a caller that edits a FROZEN A or a dual buffer behind autograd's back must call it too.  A trainable A
        is never cached (the optimiser rewrites it every step)."""
        self._cache_epoch = getattr(self, "_cache_epoch", 0) + 1
        self._sparse_checked = (None, False)
        self._off_key, self._off = None, False
        self._frozen_loss = (None, None)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)       # copies into A and the buffers through .data-like paths
        self.invalidate_caches()

    def getAlpha(self):
        # the reference computes an SVD here and discards it (:66-71); alpha = 1/d
        return torch.tensor(1. / self.in_size)

    def get_dag(self):
        return self

    def soft_thresholded_A(self):
        return 2 * (torch.sigmoid(2 * (self.A ** 2)) - .5)

    def hard_thresholded_A(self):
        if self.s_thresh:
            return self.soft_thresholded_A() * (self.soft_thresholded_A() > self.h_thresh).float()
        return self.A ** 2 * (self.A ** 2 > self.h_thresh).float()

    def _modes(self):
        """(importance mode, gate mode) from the reference's branch order (:126-153)."""
        if self.h_thresh > 0:
            imp = ops.IMP_HARD_SOFT if self.s_thresh else ops.IMP_HARD_SQ
        elif self.s_thresh:
            imp = ops.IMP_SOFT
        else:
            return ops.IMP_RAW, ops.GATE_DET
        if self.stoch_gate:
            return imp, ops.GATE_GUMBEL
        if self.noise_gate:
            return imp, ops.GATE_NOISE
        return imp, ops.GATE_DET

    def masked_inputs(self, x):
        """e: [B*d, d (+d one-hot)] -- rows are the masked copies of each sample."""
        imp, gate = self._modes()
        if gate == ops.GATE_GUMBEL and not self.gumble:
            return self._clipped_normal_gate(x)
        u1 = u2 = None
        if self.gate_noise is not None and gate != ops.GATE_DET:
            u1, u2 = self.gate_noise
        self._gate_calls += 1
        return ops.DagGateFn.apply(x, self.A, imp, gate, float(self.h_thresh), float(self.gumble_T),
                                   bool(self.hot_encoding), u1, u2, self.gate_seed, self._gate_calls)

    def _clipped_normal_gate(self, x):
        """`gumble = False` branch of the reference's stochastic_gate (:105-111): gate = relu(min(n * sigma + p + .25, 1)),
        sigma = 3 / (1 + 10 |p - .5|), n ~ N(0,1) per (b, i, j).  No driver of the reference ever clears `gumble`, so
        this branch is not fused: it is evaluated with torch ops on the device (autograd through them).
        `gate_noise = (n,)` injects the normal samples."""
        B, d = x.shape
        p = self.hard_thresholded_A() if self.h_thresh > 0 else self.soft_thresholded_A()
        n = self.gate_noise[0] if self.gate_noise is not None else torch.randn(B, d, d, device=x.device)
        sigma = 3. / (1. + 10. * torch.sqrt((p - .5) ** 2.))
        gate = torch.relu((n * sigma + p + .25).clamp_max(1.))
        e = (x.unsqueeze(1) * gate).reshape(B * d, d)
        if self.hot_encoding:
            e = torch.cat((e, torch.eye(d, device=x.device).repeat(B, 1)), 1)
        return e

    def _sparse_plan(self, x, rows, P):
        """gnf_hip.ops.SparseRows when the sparse masked-image front applies -- an MNISTCNN embedding net on the GPU, no
        gradient wanted for x or A (frozen gate), and every row of P zero outside its pixel's 5x5 window --
        else None.  rows: iterable of variable indices, None = all."""
        net = self.embedding_net
        if not self.sparse_front or self.hot_encoding or self.cond_in or not hasattr(net, "sparse_rows"):
            return None
        if not net.supports_sparse(x):
            return None
        if torch.is_grad_enabled() and (x.requires_grad or P.requires_grad):
            return None                  # the sparse kernels differentiate w.r.t. the network parameters only
        # One small reduction + a host read.  A trainable A is checked on every call (the optimiser rewrites it); a
        # frozen one (post_process) only when its storage or version counter changed, so that a training step with
        # the frozen gate has no host synchronisation in it.
        key = None if self.A.requires_grad else (self.A.data_ptr(), self.A._version, float(self.h_thresh),
                                                 bool(self.s_thresh), getattr(self, "_cache_epoch", 0))
        if key is None or key != self._sparse_checked[0]:
            if self._sparse_outside is None or self._sparse_outside.device != P.device:
                self._sparse_outside = (~ops.mnist_window_mask(P.device)).float()
            self._sparse_checked = (key, not bool((P.detach() * self._sparse_outside).count_nonzero()))
        if not self._sparse_checked[1]:
            return None
        rows = tuple(range(self.in_size)) if rows is None else tuple(int(r) for r in rows)
        plan_key = (rows, x.shape[0], x.device)
        if plan_key not in self._sparse_plans:
            if len(self._sparse_plans) > 4096:
                self._sparse_plans.clear()
            self._sparse_plans[plan_key] = ops.SparseRows(rows, x.shape[0], x.device)
        return self._sparse_plans[plan_key]

    def forward(self, x, context=None):
        no_context(context, self.cond_in)
        P = self.deterministic_importance()
        if P is not None:
            plan = self._sparse_plan(x, None, P)
            if plan is not None:
                return self.embedding_net.sparse_rows(x, P, plan)
        if hasattr(self.embedding_net, "exact_pool_ties"):
            # deterministic gate on the dense kernels (trainable A, or a gradient wanted for x): the masked copies
            # have exactly-constant regions, so the embedding net must break pool ties the way torch does
            self.embedding_net.exact_pool_ties = P is not None
        net = self.embedding_net
        if (getattr(self, "fused_front", True) and not self.hot_encoding and not self.cond_in and hasattr(net, "forward_gated")
                and x.shape[1] == self.in_size and net.supports_gated(x)):
            imp, gate = self._modes()
            if gate != ops.GATE_GUMBEL or self.gumble:
                # gate + conv front as ONE autograd node: with x frozen its backward skips the entries of dL/de that
                # dP/dA = 0 multiplies (every zero of the prior, reference :118-119)
                u1 = u2 = None
                if self.gate_noise is not None and gate != ops.GATE_DET:
                    u1, u2 = self.gate_noise
                self._gate_calls += 1
                h = net.forward_gated(x, self.A, imp, gate, float(self.h_thresh), float(self.gumble_T), u1, u2,
                                      self.gate_seed, self._gate_calls)
                return h.view(x.shape[0], self.in_size, -1)
        e = self.masked_inputs(x)
        return self.embedding_net(e).view(x.shape[0], self.in_size, -1)

    def constrainA(self, zero_threshold=.0001):
        self.A *= (self.A.clone().abs() > zero_threshold).float()
        self.A *= 1. - torch.eye(self.in_size, device=self.A.device)
        self.invalidate_caches()
        return

    def get_power_trace(self):
        """tr((I + alpha A∘A)^k) - d (:176-194).  d x d matrix power on rocBLAS through torch:
        <1% of a step and a function of the parameters only."""
        # min(1., alpha) as a tensor op: Python's min() on a device tensor forces a host<->GPU sync every step
        alpha = torch.clamp(self.alpha, max=1.) * self.alpha_factor
        if self.hutchinson != 0:
            # Hutchinson estimator of tr((I + alpha A o A)^d) (reference :179-190: `hutchinson` probe vectors e0 ~ N(0, I),
            # d = in_size matrix-vector products each, trace ~ mean e0 . (B^d e0)).  No reference driver switches it on: plain
            # torch ops on A's device, differentiable through autograd; the probes are `hutchinson_noise` ([h_iter, d], set by
            # a caller that wants to reproduce a draw) or fresh torch.randn samples.
            h_iter = int(self.hutchinson)
            B = torch.eye(self.in_size, device=self.A.device) + alpha * self.A ** 2
            noise = getattr(self, "hutchinson_noise", None)
            E0 = (noise.to(self.A.device).t() if noise is not None
                  else torch.randn(self.in_size, h_iter, device=self.A.device))          # the probes as columns
            E = E0
            for _ in range(self.in_size):
                E = B @ E
            return (E0 * E).sum() / h_iter - self.in_size
        B = (torch.eye(self.in_size, device=self.A.device) + alpha * self.A ** 2)
        return ops.PowerTraceFn.apply(B, self.exponent) - self.in_size

    def _constraints_off(self):
        """True once update_dual_param() has switched both terms off (dag_const = 0 and l1_weight = 0, reference
        :249-251): loss() is then identically 0 and its matrix power is skipped.  One host read per CHANGE of the two
        buffers (they are replaced / rewritten at epoch level only), none per step."""
        key = (id(self.dag_const), self.dag_const._version, id(self.l1_weight), self.l1_weight._version, getattr(self, "_cache_epoch", 0))
        if self._off_key != key:
            self._off_key = key
            self._off = bool(((self.dag_const == 0) & (self.l1_weight == 0)).item())
        return self._off

    def loss(self):
        """dag_const (lambd h + c/2 h^2) + l1 mean|A|, h = get_power_trace()  (reference :268-271)"""
        if not self.A.requires_grad and self._constraints_off():
            return torch.zeros((), device=self.A.device)
        if self.A.is_cuda and self.hutchinson == 0:
            if not self.A.requires_grad:
                # (a captured step bakes the value in: GraphedStep's fingerprint carries the buffers' version counters, so a
                # rewritten dual buffer means a new capture)
                # A frozen gate (post_process()) with the constraint still switched on: the term is a CONSTANT of the step --
                # no parameter receives a gradient from it -- so its d x d matrix power (6 library GEMMs + 3 launches, 0.11 ms
                # of a 2.96 ms cfg4 step after the DAG phase) is evaluated once per state of (A, the dual buffers, exponent)
                # and the same value is returned until one of them changes (version counters, as in _constraints_off)
                bufs = (self.A, self.alpha, self.lambd, self.c, self.dag_const, self.l1_weight)
                key = tuple((id(t), t.data_ptr(), t._version) for t in bufs) + (int(self.exponent), float(self.alpha_factor),
                                                                                  getattr(self, "_cache_epoch", 0))
                if self._frozen_loss[0] != key:
                    with torch.no_grad():
                        val = ops.DagLossFn.apply(self.A, self.alpha, self.alpha_factor, self.lambd, self.c, self.dag_const,
                                                  self.l1_weight, self.exponent).clone()
                    self._frozen_loss = (key, val)
                return self._frozen_loss[1]
            # one fused op: three launches around the library matrix power, buffers read on the device
            return ops.DagLossFn.apply(self.A, self.alpha, self.alpha_factor, self.lambd, self.c, self.dag_const,
                                       self.l1_weight, self.exponent)
        lag_const = self.get_power_trace()
        return self.dag_const * (self.lambd * lag_const + self.c / 2 * lag_const ** 2) \
            + self.l1_weight * self.A.abs().mean()

    def post_process(self, zero_threshold=None):
        def adj(th):
            return (self.soft_thresholded_A().data.clone().abs() > th).float().detach().cpu().numpy()
        if zero_threshold is None:
            zero_threshold = .1
            while not nx.is_directed_acyclic_graph(_digraph(adj(zero_threshold))):
                zero_threshold += .05
        self.stoch_gate = False
        self.noise_gate = False
        self.s_thresh = False
        self.h_thresh = 0.
        # written INTO A's storage (the reference rebinds A.data, :88): A may be a view of a flat parameter buffer
        # (gnf_hip.dp.FlatState) and of a captured hipGraph, both of which must keep seeing it
        with torch.no_grad():
            self.A.copy_((self.soft_thresholded_A().abs() > zero_threshold).float()
                         * (1. - torch.eye(self.in_size, device=self.A.device)))
        self.A.requires_grad = False
        self.A.grad = None
        self.invalidate_caches()

    def _reopen(self, A=None):
        self.stoch_gate = True
        self.noise_gate = False
        self.s_thresh = True
        self.h_thresh = 0.
        if A is not None:
            # the reference installs a NEW nn.Parameter here (:224), which its optimiser never sees (it keeps stepping
            # the old object): A silently stops training after a failed post-processing.  Here the saved values go back
            # into the same storage, so the optimiser (flat buffer or torch.optim) keeps updating A.
            with torch.no_grad():
                self.A.copy_(A)
        self.A.requires_grad = True
        self.A.grad = None               # the reference parks A.clone() here until its next zero_grad() (:226,244)
        self.invalidate_caches()
        self._set("alpha", self.getAlpha())
        self._set("prev_trace", self.get_power_trace().detach())

    def _set(self, name, value):
        """buffer <- value IN PLACE (the reference rebinds new tensors, e.g. `self.lambd = self.lambd + ...`): device
        addresses stay valid for captured hipGraphs, and the value lands on A's device (the reference leaves CPU scalars
        `torch.tensor(1.)` behind, :231,233-234)."""
        buf = getattr(self, name)
        with torch.no_grad():
            buf.copy_(torch.as_tensor(value, dtype=buf.dtype, device=buf.device))
        self.invalidate_caches()

    def update_dual_param(self):
        """Augmented-Lagrangian update of (lambd, c) / post-processing (:196-260)."""
        self.invalidate_caches()             # epoch-level decisions are taken on freshly evaluated values
        with torch.no_grad():
            lag_const = self.get_power_trace()
            while self.dag_const > 0. and lag_const < self.tol and self.exponent < self.in_size:
                self.exponent += 50
                lag_const = self.get_power_trace()
            if self.dag_const > 0. and lag_const > self.tol:
                self._set("lambd", self.lambd + self.c * lag_const)
                if lag_const.abs() > self.gamma * self.prev_trace.abs():
                    self.c *= self.eta
                self._set("prev_trace", lag_const)
            elif self.dag_const > 0.:
                A_before = self.A.detach().clone()
                self.post_process()
                self._set("alpha", self.getAlpha())
                lag_const = self.get_power_trace()
                if lag_const > 0.:
                    self._reopen(A_before)
                    self.c *= 1 / self.eta
                    self._set("lambd", self.lambd + self.c * lag_const)
                    self._set("dag_const", 1.)
                else:
                    self._set("dag_const", 0.)
                    self._set("l1_weight", 0.)
            else:
                G = _digraph(self.A.detach().cpu().numpy() ** 2)
                try:
                    nx.find_cycle(G)
                    self._reopen()
                    self._set("dag_const", 1.)
                except nx.NetworkXNoCycle:
                    self.is_invertible = True
        return lag_const

    def depth(self):
        G = _digraph((self.A.detach() > 0).float().cpu().numpy())
        if self.is_invertible or nx.is_directed_acyclic_graph(G):
            return int(nx.dag_longest_path_length(G))
        return 0

    def levels(self, P=None, with_host=False):
        """Topological generations of the dependency graph (edge j -> i where the importance P[i, j] != 0; default: the
        graph depth() measures, A[i, j] > 0): level k holds the variables whose longest parent chain has length k.
        None if the graph has a cycle."""
        adj = ((self.A.detach() > 0) if P is None else (P.detach() != 0)).cpu().numpy()
        d = adj.shape[0]
        indeg = adj.sum(1).astype(np.int64)                 # number of parents of i
        children = [np.nonzero(adj[:, j])[0] for j in range(d)]
        frontier = np.nonzero(indeg == 0)[0]
        out, done = [], 0
        while frontier.size:
            rows = torch.as_tensor(frontier, dtype=torch.long, device=self.A.device)
            out.append((rows, tuple(int(r) for r in frontier)) if with_host else rows)   # host copy: no read-back later
            done += frontier.size
            nxt = []
            for j in frontier:
                for i in children[j]:
                    indeg[i] -= 1
                    if indeg[i] == 0:
                        nxt.append(i)
            frontier = np.array(sorted(nxt), dtype=np.int64)
        return out if done == d else None

    def deterministic_importance(self):
        """The matrix the deterministic branches of forward multiply x with (reference :126-153), or None when a
        stochastic / noisy gate is active."""
        if self.stoch_gate or self.noise_gate:
            if self.h_thresh > 0 or self.s_thresh:
                return None
        if self.h_thresh > 0:
            return self.hard_thresholded_A()
        if self.s_thresh:
            return self.soft_thresholded_A()
        return self.A

    def forward_rows(self, x, rows, P, host_rows=None, variable_major=False):
        """h[:, rows, :] only: the conditioner output of row i depends on x through x * P[i] alone, so a
        level-scheduled inversion evaluates each row exactly once.  variable_major: the result as
        [R, B, out] (any strides) instead of [B, R, out] -- the sparse kernels produce that layout, and the level loop
        of the inversion consumes it without a permuting copy."""
        B, R = x.shape[0], rows.numel()
        plan = self._sparse_plan(x, rows.tolist() if host_rows is None else host_rows, P)
        if plan is not None:
            return self.embedding_net.sparse_rows(x, P, plan, variable_major=variable_major)
        if variable_major:
            return self.forward_rows(x, rows, P, host_rows).permute(1, 0, 2)
        if hasattr(self.embedding_net, "exact_pool_ties"):
            self.embedding_net.exact_pool_ties = True                    # deterministic gate by construction
        e = x.unsqueeze(1) * P[rows].unsqueeze(0)                        # [B, R, d]
        if self.hot_encoding:
            hot = torch.zeros(R, self.in_size, device=x.device, dtype=x.dtype)
            hot[torch.arange(R, device=x.device), rows] = 1.
            e = torch.cat((e, hot.unsqueeze(0).expand(B, -1, -1)), 2)
        return self.embedding_net(e.reshape(B * R, -1)).view(B, R, -1)

    def step(self, epoch_number, loss_avg=0.):
        """Once per epoch (:273-293): exponent back-off and dual update schedule."""
        self.invalidate_caches()             # whatever was written since the last epoch, however: decide on fresh values
        with torch.no_grad():
            lag_const = self.get_power_trace()
            if lag_const > 50:
                self.exponent -= 5
                self.exponent = self.exponent if self.exponent > 3 else 3
            if epoch_number % self.nb_epoch_update == 0 and epoch_number > 0:
                if self.loss().abs() < abs(loss_avg) / 2 or self.no_update > 10:
                    self.update_dual_param()
                    self.no_update = 0
                else:
                    self.no_update += 1
