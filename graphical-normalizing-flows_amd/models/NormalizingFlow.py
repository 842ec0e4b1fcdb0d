"""Flow composition on the MI355X path: one (conditioner, normalizer) step, a stack of steps, and the multi-scale
image flow.  Class and method names, constructor arguments and `state_dict` keys are those of the reference's
models/NormalizingFlow.py (the drivers call getConditioners / getNormalizers / DAGness / step / isInvertible / invert
by name); the bodies are written against gnf_hip.ops.

    step:   h = conditioner(x);  z, jac = normalizer(x, h);  log|det J| = sum_i log jac_i      (reference :61-70)
    stack:  steps in sequence, feature order reversed between steps, log-dets added             (reference :110-126)
    loss:   sum of conditioner constraint terms - mean(log|det J| + log N(z))                  (reference :128-146)
"""
import weakref

import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditionners import Conditioner, DAGConditioner
from .Normalizers import Normalizer


# captured inversion graphs per NormalizingFlowStep, kept OUTSIDE the modules: a hipGraph is neither copyable nor
# picklable, and flows are deep-copied / saved by the drivers
_INV_GRAPHS = weakref.WeakKeyDictionary()


def _is_dag(conditioner):
    # exact type test, as everywhere in the reference (subclasses are deliberately not recognised)
    return type(conditioner) is DAGConditioner


class NormalizingFlow(nn.Module):
    """Protocol shared by all flows.  forward(x, context=None) -> (z, log|det J|).  The aggregate queries have
    default implementations in terms of getConditioners(), which every concrete flow provides."""

    def __init__(self):
        super().__init__()

    def forward(self, x, context=None):
        raise NotImplementedError

    def invert(self, z, context=None):
        raise NotImplementedError

    def getConditioners(self):
        raise NotImplementedError

    def getNormalizers(self):
        raise NotImplementedError

    def constraintsLoss(self):
        total = None                                    # (sum(..., 0.) of the reference: 0. + loss is one more launch)
        for c in self.getConditioners():
            if _is_dag(c):
                total = c.loss() if total is None else total + c.loss()
        return 0. if total is None else total

    def DAGness(self):
        return [c.get_power_trace() if _is_dag(c) else 0. for c in self.getConditioners()]

    def step(self, epoch_number, loss_avg):
        for c in self.getConditioners():
            if _is_dag(c):
                c.step(epoch_number, loss_avg)

    def isInvertible(self):
        return all(c.is_invertible for c in self.getConditioners())


class NormalizingFlowStep(NormalizingFlow):
    """One autoregressive-style transformation.  A normalizer that offers `forward_logdet` returns the row-summed
    log-Jacobian itself: the Affine normalizer from the kernel that computes z (log-det and Normal log-density reduced
    in the same pass), the Monotonic one from a single fused reduction pass over its (z, jac) (its integrand kernels
    spread a row over many wavefronts); any other Normalizer plug-in goes through the log + row-sum reduction kernel."""

    def __init__(self, conditioner: Conditioner, normalizer: Normalizer):
        super().__init__()
        self.conditioner = conditioner
        self.normalizer = normalizer
        self.level_schedule = True       # invert(): topological level schedule for DAG conditioners
        self.graph_invert = True         # ... replayed as a hipGraph from the second pass of a shape on (frozen A, GPU)

    def getConditioners(self):
        return [self.conditioner]

    def getNormalizers(self):
        return [self.normalizer]

    def forward(self, x, context=None):
        h = self.conditioner(x, context)
        fused = getattr(self.normalizer, "forward_logdet", None)
        if fused is not None:
            return fused(x, h, context)
        z, jac = self.normalizer(x, h, context)
        return z, ops.LogSumRowsFn.apply(jac)

    # -- inversion ----------------------------------------------------------------------------------------------
    def _levels(self, importance):
        """the level schedule of the current gate, recomputed only when A (storage or version counter) or the gate's
        thresholds changed: DAGConditioner.levels() copies the 784 x 784 adjacency to the host and walks it in Python
        (a few ms per call, a host synchronisation in front of every sampling pass otherwise)"""
        cond = self.conditioner
        # A trainable A is rewritten by the optimiser through a raw pointer (gnf_hip.dp: fused Adam on the flat buffer, a
        # replayed hipGraph), which moves neither its version counter nor its address: nothing is cached for it.  A frozen
        # one is keyed on (storage, version, thresholds, the conditioner's cache epoch -- see invalidate_caches()).
        key = (None if cond.A.requires_grad else
               (cond.A.data_ptr(), cond.A._version, float(cond.h_thresh), bool(cond.s_thresh), cond.A.device,
                getattr(cond, "_cache_epoch", 0)))
        if key is None or getattr(self, "_levels_key", None) != key:
            lv = cond.levels(importance, with_host=True)
            if lv is not None and cond.A.shape[0] == 784:
                # the order of the variables INSIDE a level is free: take the one the sparse embedding kernels work in (crop
                # origin, then pixel), so that their output needs no un-sorting gather
                lv = [(torch.as_tensor(sorted(hr, key=lambda r: (8 * ops.crop_origin(r // 28) + ops.crop_origin(r % 28), r)),
                                       dtype=torch.long, device=rows.device),
                       tuple(sorted(hr, key=lambda r: (8 * ops.crop_origin(r // 28) + ops.crop_origin(r % 28), r))))
                      for rows, hr in lv]
            self._levels_val = lv
            self._levels_all = torch.cat([rows for rows, _ in lv]) if lv else None   # one gather of z per pass
            self._levels_rows32 = [rows.to(torch.int32) for rows, _ in lv] if lv else None   # scatter tables of the inverse
            self._levels_key = key
            _INV_GRAPHS.pop(self, None)                 # graphs captured for another gate replay another schedule
        return self._levels_val, key

    def _invert_levels_body(self, z, levels, importance, context):
        """variable-major inside: z is gathered ONCE into [sum of level sizes, B] (a level is a contiguous slice), the
        conditioner rows come as [R, B, out] -- the layout the sparse kernels write -- and the normalizer's inverse is
        element-wise in whatever two leading dimensions z and h share.  Per level that leaves the embedding front, the
        inverse and one scatter into x (no gather of z, no un-sorting or permuting copy of h)."""
        cond = self.conditioner
        x = torch.zeros_like(z)
        zt = z.t()[self._levels_all]                    # [sum R, B], contiguous
        off = 0
        into = getattr(self.normalizer, "inverse_transform_into", None) if context is None else None
        rows32 = getattr(self, "_levels_rows32", None)
        if into is not None and (rows32 is None or len(rows32) != len(levels)):
            into = None
        for k, (rows, host_rows) in enumerate(levels):
            R = rows.numel()
            h = cond.forward_rows(x, rows, importance, host_rows, variable_major=True)
            # the normalizer writes its [R, B] result into columns `rows` of x itself where it can
            if into is None or not into(zt[off:off + R], h, x, rows32[k]):
                x[:, rows] = self.normalizer.inverse_transform(zt[off:off + R], h, context).t()
            off += R
        return x

    GRAPH_INVERT_MAX = 8                 # captured (batch shape, node count) variants kept per step

    def _holders(self):
        """objects that can build parameter-only images ONCE for a whole inversion (attribute name, context factory):
        the normalizer's packed integrand weights, the embedding net's sparse-front tables"""
        out = []
        if getattr(self.normalizer, "hold_pack", None) is not None:
            out.append((self.normalizer, "_held_pack", self.normalizer.hold_pack))
        net = getattr(self.conditioner, "embedding_net", None)
        if getattr(net, "hold_prepared", None) is not None:
            out.append((net, "_held_prep", net.hold_prepared))
        return out

    def _invert_by_levels(self, z, context):
        """DAG conditioner with a deterministic gate: every variable is inverted once, after its parents -- d
        conditioner rows in total instead of (depth + 1) * d.  Same values as the fixed-point passes (non-parents are
        masked by exact zeros either way).  Returns None when not applicable.

        On the GPU the pass of a given (gate, batch shape, node count) is captured into a hipGraph the second time it is
        asked for and replayed from then on (`graph_invert`): with a frozen A the schedule, the row tables and the sparse
        plans are static, and a sampling pass over MNIST is ~650 launches of 5-100 us (reference ImageExperiments.py:341-350
        samples after post_process()).  The weight image of the normalizer is packed INSIDE the graph, so a replay reads
        the current parameters."""
        cond = self.conditioner
        if not (_is_dag(cond) and context is None and self.level_schedule):
            return None
        importance = cond.deterministic_importance()
        if importance is None:
            return None
        levels, lkey = self._levels(importance)
        if levels is None:
            return None
        # a captured pass bakes the ADDRESS of the importance matrix in: only A itself qualifies -- the thresholded forms
        # (s_thresh / h_thresh > 0, the state the reference's threshold sweep sets) are temporaries freed after this call
        graphable = (self.graph_invert and z.is_cuda and not cond.A.requires_grad and z.dtype == torch.float32
                     and importance.data_ptr() == cond.A.data_ptr()
                     and len(levels) > 4 and not torch.cuda.is_current_stream_capturing())
        if not graphable:
            return self._invert_levels_body(z, levels, importance, context)
        # what a captured pass bakes in: shapes, the node count, the front, and the ADDRESSES of every parameter and buffer
        # (values may change freely; a re-bound or moved tensor must not be read through a stale pointer)
        key = (tuple(z.shape), int(getattr(self.normalizer, "nb_steps", 0)), bool(getattr(cond, "sparse_front", False)),
               tuple(t.data_ptr() for t in self.parameters()), tuple(t.data_ptr() for t in self.buffers()))
        graphs = _INV_GRAPHS.setdefault(self, {})
        entry = graphs.get(key)
        if entry is None:                               # first request: eager (workspaces, plans and tables come to exist)
            graphs[key] = "warm"
            return self._invert_levels_body(z, levels, importance, context)
        if entry == "eager":                            # a capture of this variant failed once: launch by launch from then on
            return self._invert_levels_body(z, levels, importance, context)
        if entry == "warm":
            import gc
            zbuf = z.clone()
            import contextlib
            holders = self._holders()
            saved = [(obj, attr, getattr(obj, attr)) for obj, attr, _ in holders]
            for obj, attr, _ in holders:                # the build launches belong INTO the graph (current parameters on replay)
                setattr(obj, attr, None)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # once on the capture stream's allocator pool
                with contextlib.ExitStack() as stack:
                    for _, _, make in holders:
                        stack.enter_context(make())
                    self._invert_levels_body(zbuf, levels, importance, context)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()                                # no finaliser may run inside the capture
            failed = None
            try:
                with torch.cuda.graph(graph), contextlib.ExitStack() as stack:
                    for _, _, make in holders:
                        stack.enter_context(make())
                    xbuf = self._invert_levels_body(zbuf, levels, importance, context)
            except Exception as exc:                    # noqa: BLE001  (e.g. a host synchronisation inside a user-supplied
                failed = exc                            # integrand / embedding module): this variant stays eager
            finally:
                if gc_was_on:
                    gc.enable()
                for obj, attr, val in saved:
                    setattr(obj, attr, val)
            if failed is not None:
                import warnings
                warnings.warn("NormalizingFlowStep.invert: hipGraph capture of the level schedule failed (%r); this variant "
                              "runs eagerly from now on" % (failed,), RuntimeWarning, stacklevel=2)
                graphs[key] = "eager"
                torch.cuda.synchronize()
                return self._invert_levels_body(z, levels, importance, context)
            if len(graphs) > self.GRAPH_INVERT_MAX:
                graphs.clear()
            entry = graphs[key] = (graph, zbuf, xbuf)
        graph, zbuf, xbuf = entry
        zbuf.copy_(z, non_blocking=True)
        graph.replay()
        return xbuf.clone()

    def invert(self, z, context=None):
        """Reference :98-107: fixed point of x <- normalizer^-1(z, conditioner(x)) from x = 0, depth()+1 passes, early
        exit once a pass changes nothing (its progress print is dropped)."""
        import contextlib
        with torch.no_grad(), contextlib.ExitStack() as stack:   # parameters do not change inside: build their images once
            for _, _, make in self._holders():
                stack.enter_context(make())
            x = self._invert_by_levels(z, context)
            if x is not None:
                return x
            x = torch.zeros_like(z)
            for _ in range(self.conditioner.depth() + 1):
                previous = x
                x = self.normalizer.inverse_transform(z, self.conditioner(x, context), context)
                if torch.equal(x, previous):
                    break
        return x


class FCNormalizingFlow(NormalizingFlow):
    """Steps applied in sequence on flat [B, d] data."""

    def __init__(self, steps, z_log_density):
        super().__init__()
        self.steps = nn.ModuleList(steps)
        self.z_log_density = z_log_density

    def getConditioners(self):
        return [c for s in self.steps for c in s.getConditioners()]

    def getNormalizers(self):
        return [n for s in self.steps for n in s.getNormalizers()]

    def forward(self, x, context=None):
        logdet = None
        last = len(self.steps) - 1
        for k, flow_step in enumerate(self.steps):
            z, ld = flow_step(x, context)
            logdet = ld if logdet is None else logdet + ld      # (the reference's 0. + ld is one more launch)
            if k < last:
                x = torch.flip(z, dims=[1])      # the reference's z[:, inv_idx] (:120-123)
        return z, logdet                         # the last step's z is returned un-flipped (:126)

    def loss(self, z, jac):
        dens = self.z_log_density
        fused = (z.is_cuda and jac.is_cuda and jac.dim() == 1 and z.dim() == 2 and jac.shape[0] == z.shape[0]
                 and z.dtype == torch.float32 and jac.dtype == torch.float32)
        c = self.constraintsLoss()
        plain_c = isinstance(c, float) and c == 0.
        dev_c = torch.is_tensor(c) and c.is_cuda and c.dim() == 0 and c.dtype == torch.float32
        # the fold applies to the factories' base density and to nothing that merely inherits from it: a subclass that
        # overrides forward (tempered / scaled / conditional density) is called as the reference calls it
        std_normal = (getattr(dens, "standard_normal", False)
                      and getattr(type(dens), "_std_forward", None) is type(dens).forward)
        if fused and std_normal and ops.nll_loss_fits(z) and (plain_c or dev_c):
            # constraints - mean(log|det J| + log N(z)) in ONE launch that reads z itself (the standard-normal base density of
            # the factories; any other z_log_density module is called as the reference calls it)
            return ops.NllLossFn.apply(z, jac, None if plain_c else c)
        logn = dens(z)
        if fused and logn.shape == jac.shape and jac.shape[0] > 0 and logn.dtype == torch.float32:
            if plain_c:
                return ops.NllMeanFn.apply(jac, logn)                            # -(jac + logn).mean(), one launch
            if dev_c:
                return ops.NllMeanFn.apply(jac, logn, c)                         # constraints - mean(...), the same launch
            return c + ops.NllMeanFn.apply(jac, logn)
        return c - (jac + logn).mean()

    def invert(self, z, context=None):
        """Exact inverse of forward for any number of steps.  The reference (:166-169) visits steps[-0] == steps[0]
        first and never undoes the flip, so it only inverts nb_flow = 1 flows; for nb_flow = 1 this is
        identical to it."""
        for k in range(len(self.steps) - 1, -1, -1):
            x = self.steps[k].invert(z, context)
            z = torch.flip(x, dims=[1]) if k else x
        return z


class CNNormalizingFlow(FCNormalizingFlow):
    """Multi-scale image flow (reference :172-226): after every scale the image is cut into d_c x d_h x d_w blocks;
    the first element of each block goes on to the next (coarser) flow, the others are emitted as latent variables."""

    def __init__(self, steps, z_log_density, dropping_factors):
        super().__init__(steps, z_log_density)
        self.dropping_factors = dropping_factors

    @staticmethod
    def _kept_shape(img_size, drop):
        return tuple(int(n / f) for n, f in zip(img_size, drop))

    @staticmethod
    def _blocks(z, img_size, drop):
        """[B, C*H*W] -> [B, c, h, w, d_c*d_h*d_w]  (equals the unfold chain of reference :184-185)"""
        c, h, w = CNNormalizingFlow._kept_shape(img_size, drop)
        d_c, d_h, d_w = drop
        return z.view(-1, c, d_c, h, d_h, w, d_w).permute(0, 1, 3, 5, 2, 4, 6).reshape(z.shape[0], c, h, w, -1)

    def forward(self, x, context=None):
        batch = x.shape[0]
        logdet, latents = None, []
        for flow, drop in zip(self.steps, self.dropping_factors):
            z, ld = flow(x, context)
            logdet = ld if logdet is None else logdet + ld
            blocks = self._blocks(z, flow.img_sizes, drop)
            latents.append(blocks[..., 1:].reshape(batch, -1))
            x = blocks[..., 0].reshape(batch, -1)
        latents.append(x)
        return torch.cat(latents, 1), logdet

    def invert(self, z, context=None):
        batch = z.shape[0]
        # slice z back into the per-scale latent blocks, in emission order
        parts, start = [], 0
        for flow, drop in zip(self.steps, self.dropping_factors):
            full = flow.img_sizes[0] * flow.img_sizes[1] * flow.img_sizes[2]
            c, h, w = self._kept_shape(flow.img_sizes, drop)
            width = full - c * h * w if full != c * h * w else full
            parts.append(z[:, start:start + width])
            start += width
        x = None
        for flow, drop, zk in zip(reversed(self.steps), reversed(self.dropping_factors), reversed(parts)):
            c, h, w = self._kept_shape(flow.img_sizes, drop)
            if c * h * w != flow.img_sizes[0] * flow.img_sizes[1] * flow.img_sizes[2]:
                # re-interleave the element kept for the coarser scale with the dropped ones of each block
                blocks = torch.cat((x.view(batch, c, h, w, 1), zk.reshape(batch, c, h, w, -1)), 4)
                zk = blocks.view(batch, c, h, w, *drop).permute(0, 1, 4, 2, 5, 3, 6).reshape(batch, -1)
            x = flow.invert(zk.reshape(batch, -1), context)
        return x
