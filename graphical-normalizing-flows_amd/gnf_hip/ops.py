"""torch.autograd wrappers around the C-ABI kernels (gnf_hip.abi).

PyTorch's role here is plumbing: tensor allocation, autograd graph bookkeeping and the
current HIP stream.  Every numerical step of the flow hot path runs in libgnf_hip.so."""
import ctypes
import math
import os

import weakref

import numpy as np

import torch

from . import abi
from .abi import ptr, stream, call


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=like.device)


# ----------------------------------------------------------------------------- gradient slots
# A training state that keeps every gradient in ONE flat buffer (dp.FlatState: one all-reduce, one Adam launch) registers
# the slot of each parameter here; a backward that produces a parameter gradient writes it straight into the slot instead
# of a fresh tensor, and the state's gradient pack finds it already in place (no concatenation launch, 12.8 MB less
# traffic per MADE step).  A slot is handed out once per pack: a parameter used twice in a graph, or a second micro-batch
# accumulating into `.grad`, gets a fresh tensor and autograd adds it as usual.
_grad_slots = {}                       # parameter data_ptr -> (weakref to the owning state, index into its params)


def register_grad_slots(owner, params):
    """owner: object with `grad_views` (one tensor per parameter, same numel) and a set `_taken`"""
    ref = weakref.ref(owner)
    for i, p in enumerate(params):
        _grad_slots[p.data_ptr()] = (ref, i)


def unregister_grad_slots(owner):
    """drop the slots of `owner` (dp.FlatState.__del__ / close).  Lookups are safe without it -- a parameter of a live state
    is a view of the state's own flat buffer, so its address cannot be handed to anybody else, and an entry whose owner has
    died is deleted by the lookup that finds it -- but a dead state should not leave entries behind."""
    for k in [k for k, (ref, _) in _grad_slots.items() if ref() is owner or ref() is None]:
        del _grad_slots[k]


def grad_out(p):
    """the tensor a backward writes the gradient of parameter (or saved alias of it) `p` into"""
    ent = _grad_slots.get(p.data_ptr())
    if ent is not None:
        owner = ent[0]()
        if owner is None:
            del _grad_slots[p.data_ptr()]
        else:
            i = ent[1]
            v = owner.grad_views[i]
            if (i not in owner._taken and v.numel() == p.numel() and p.is_contiguous() and v.device == p.device
                    and v.dtype == p.dtype):
                owner._taken.add(i)
                return v.view(p.shape)
    return torch.empty_like(p)


_open_slots = set()                    # (id(owner), index) of shared slots taken in the backward pass that is running NOW


def _close_slots():
    _open_slots.clear()


def grad_out_shared(p, accumulate_ok=False):
    """(tensor, finish, accumulate) for a parameter that receives SEVERAL gradient contributions per backward (the DAG matrix A: gate
    and acyclicity term).  The first contribution takes the flat-buffer slot as grad_out does; a later one OF THE SAME
    BACKWARD PASS is written to a scratch tensor and finish() adds it INTO the slot and returns None, so autograd sees one
    contribution -- the slot -- instead of summing two tensors into a third that the gradient pack then copies (add +
    2.4 MB copy per cfg4 step).  "Same pass" is tracked by a callback the autograd engine runs when the pass ends: a slot
    left taken by an earlier pass (a backward without a gradient pack behind it, a previous micro-batch) is never added
    into -- that contribution gets a fresh tensor and autograd sums as usual.  finish(t) returns what the backward hands
    to autograd.  accumulate_ok: the caller's kernel can add into its output; a later contribution then gets the slot
    itself with accumulate = True (no scratch tensor, no add launch)."""
    ent = _grad_slots.get(p.data_ptr())
    owner = ent[0]() if ent is not None else None
    if owner is not None:
        i = ent[1]
        v = owner.grad_views[i]
        if v.numel() == p.numel() and p.is_contiguous() and v.device == p.device and v.dtype == p.dtype:
            key = (id(owner), i)
            if i not in owner._taken:
                owner._taken.add(i)
                if not _open_slots:
                    torch.autograd.Variable._execution_engine.queue_callback(_close_slots)
                _open_slots.add(key)
                return v.view(p.shape), (lambda t: t), False
            if key in _open_slots:
                if accumulate_ok:                                   # the caller's kernel adds INTO the slot itself
                    return v.view(p.shape), (lambda t: None), True

                def finish(t, v=v):
                    v.add_(t.reshape(-1))
                    return None
                return torch.empty_like(p), finish, False
    return torch.empty_like(p), (lambda t: t), False


# ----------------------------------------------------------------------------- Affine normalizer
class AffineFn(torch.autograd.Function):
    """(z, jac, logdet, logn) of models/Normalizers/AffineNormalizer.py:9-12 fused with the log|det J| row reduction of
    models/NormalizingFlow.py:70 and (want_logn) the Normal log-density of z (NormalizingFlowFactories.py:15-16): the
    pass that produces z also reduces it, and the backward recomputes z instead of reading it."""

    @staticmethod
    def forward(ctx, x, h, clamp_inplace=False, want_jac=True, want_logn=False):
        x = x.contiguous()
        B, d = x.shape
        ctx.set_materialize_grads(False)     # a step that only uses logdet / logn must not get zero tensors for z and jac
        z, logdet = _empty((B, d), x), _empty((B,), x)
        jac = _empty((B, d), x) if want_jac else None       # the fused step only needs log|det J|: 4 B/elem less traffic
        logn = _empty((B,), x) if want_logn else None
        h_bwd = h.detach().clone() if clamp_inplace else h
        call("gnf_affine_fwd", ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2), ptr(z), ptr(jac), ptr(logdet),
             ptr(logn), int(bool(clamp_inplace)), B, d, stream())
        ctx.save_for_backward(x, h_bwd)
        ctx.hshape = tuple(h.shape)
        if jac is None:
            jac = z.new_empty(0)
            ctx.mark_non_differentiable(jac)
        if logn is None:
            logn = z.new_empty(0)
            ctx.mark_non_differentiable(logn)
        return z, jac, logdet, logn

    @staticmethod
    def backward(ctx, gz, gjac, glogdet, glogn=None):
        x, h = ctx.saved_tensors
        B, d = x.shape
        hs = ctx.hshape[2]
        # same memory format as h so that e.g. MADE's permuted view flows back without a copy
        gh = torch.empty_like(h) if hs == 2 else torch.zeros_like(h)
        gx = _empty((B, d), x) if ctx.needs_input_grad[0] else None
        call("gnf_affine_bwd", ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2),
             ptr(gz.contiguous()) if gz is not None else None,
             ptr(gjac.contiguous()) if (gjac is not None and gjac.numel() > 0) else None,
             ptr(glogdet.contiguous()) if glogdet is not None else None,
             ptr(glogn.contiguous()) if (glogn is not None and glogn.numel() > 0) else None,
             ptr(gx), ptr(gh), gh.stride(0), gh.stride(1), gh.stride(2), B, d, stream())
        return gx, gh, None, None, None


def affine_inverse(z, h):
    """(z - mu)/sigma   (AffineNormalizer.py:14-17)."""
    z = z.contiguous()
    B, d = z.shape
    x = _empty((B, d), z)
    call("gnf_affine_inv", ptr(z), ptr(h), h.stride(0), h.stride(1), h.stride(2), ptr(x), B, d, stream())
    return x


# ----------------------------------------------------------------------------- row reductions
class LogSumRowsFn(torch.autograd.Function):
    """torch.log(jac).sum(1)   (models/NormalizingFlow.py:70)."""

    @staticmethod
    def forward(ctx, jac):
        jac = jac.contiguous()
        B, d = jac.shape
        out = _empty((B,), jac)
        call("gnf_logsum_rows_fwd", ptr(jac), ptr(out), B, d, stream())
        ctx.save_for_backward(jac)
        return out

    @staticmethod
    def backward(ctx, g):
        jac, = ctx.saved_tensors
        B, d = jac.shape
        gj = _empty((B, d), jac)
        call("gnf_logsum_rows_bwd", ptr(jac), ptr(g.contiguous()), ptr(gj), B, d, stream())
        return gj


class NormalLogDensityFn(torch.autograd.Function):
    """-.5*(log(2 pi) + z**2).sum(1)   (models/NormalizingFlowFactories.py:15-16)."""

    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        B, d = z.shape
        out = _empty((B,), z)
        call("gnf_normal_logdensity_fwd", ptr(z), ptr(out), B, d, stream())
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        z, = ctx.saved_tensors
        B, d = z.shape
        gz = _empty((B, d), z)
        call("gnf_normal_logdensity_bwd", ptr(z), ptr(g.contiguous()), ptr(gz), B, d, stream())
        return gz


class NllReduceFn(torch.autograd.Function):
    """(log(jac).sum(1), -.5*(log(2 pi) + z**2).sum(1)): the two row reductions of a flow step's tail
    (models/NormalizingFlow.py:70, NormalizingFlowFactories.py:15-16) in one pass over z and jac, and one backward
    launch producing both cotangents (the Monotonic normalizer's kernels cannot reduce over a row themselves)."""

    @staticmethod
    def forward(ctx, z, jac):
        z, jac = z.contiguous(), jac.contiguous()
        B, d = z.shape
        ctx.set_materialize_grads(False)
        logdet, logn = _empty((B,), z), _empty((B,), z)
        call("gnf_nll_reduce_fwd", ptr(z), ptr(jac), ptr(logdet), ptr(logn), B, d, stream())
        ctx.save_for_backward(z, jac)
        return logdet, logn

    @staticmethod
    def backward(ctx, glogdet, glogn):
        z, jac = ctx.saved_tensors
        B, d = z.shape
        gz, gjac = _empty((B, d), z), _empty((B, d), z)
        call("gnf_nll_reduce_bwd", ptr(z), ptr(jac), ptr(glogdet.contiguous()) if glogdet is not None else None,
             ptr(glogn.contiguous()) if glogn is not None else None, None, ptr(gz), ptr(gjac), B, d, stream())
        return gz, gjac


class NllMeanFn(torch.autograd.Function):
    """addend - (logdet + logn).mean(): FCNormalizingFlow.loss (models/NormalizingFlow.py:144-146; addend = the constraints
    term as a 0-dim tensor, or None) and its cotangents, one launch each way."""

    @staticmethod
    def forward(ctx, logdet, logn, addend=None):
        logdet, logn = logdet.contiguous(), logn.contiguous()
        out = _empty((), logdet)
        call("gnf_nll_mean_fwd", ptr(logdet), ptr(logn), ptr(addend), ptr(out), logdet.shape[0], stream())
        ctx.B = logdet.shape[0]
        ctx.like = logdet
        return out

    @staticmethod
    def backward(ctx, g):
        gl, gn = _empty((ctx.B,), ctx.like), _empty((ctx.B,), ctx.like)
        call("gnf_nll_mean_bwd", ptr(g.contiguous()), ptr(gl), ptr(gn), ctx.B, stream())
        return gl, gn, (g if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] else None)


class NllLossFn(torch.autograd.Function):
    """addend - (logdet + logN(z)).mean() with the Normal log-density computed from z ITSELF inside the loss launch
    (FCNormalizingFlow.loss, models/NormalizingFlow.py:144-146 with NormalizingFlowFactories.py:15-16): one launch each
    way.  Until round 4 the density was reduced by the kernel that produced z and remembered on the tensor; a z rewritten
    through `.data` (no version bump) was then scored with the old density.  Now the loss reads the z it is handed."""

    @staticmethod
    def forward(ctx, z, logdet, addend=None):
        z, logdet = z.contiguous(), logdet.contiguous()
        B, d = z.shape
        out = _empty((), z)
        call("gnf_nll_loss_fwd", ptr(z), ptr(logdet), ptr(addend), ptr(out), B, d, stream())
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        z, = ctx.saved_tensors
        B, d = z.shape
        gz, gl = _empty((B, d), z), _empty((B,), z)
        call("gnf_nll_loss_bwd", ptr(g.contiguous()), ptr(z), ptr(gz), ptr(gl), B, d, stream())
        return gz, gl, (g if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] else None)


_LOSS_MAX = None


def nll_loss_fits(z):
    """True when NllLossFn takes this z (one workgroup reads all of it: up to 2^20 elements)"""
    global _LOSS_MAX
    if _LOSS_MAX is None:
        _LOSS_MAX = int(abi.load().gnf_nll_loss_max_elems())
    return z.dim() == 2 and 0 < z.numel() <= _LOSS_MAX


def colsum(a):
    """sum over rows of a contiguous [M,N] tensor (bias gradients)."""
    M, N = a.shape
    out = _empty((N,), a)
    ws = _ws(abi.load().gnf_colsum_ws_bytes(M, N), a)
    call("gnf_colsum", ptr(a), N, ptr(out), M, N, ptr(ws), stream())
    return out


# ----------------------------------------------------------------------------- MFMA GEMM / MLP chains
def gemm(A, a_strides, Bm, b_strides, C, c_strides, M, N, K, Bmask=None, bias=None, Cmask=None, cm_strides=(0, 0),
         gate=None, g_strides=(0, 0), relu=False):
    nws = abi.load().gnf_gemm_ws_bytes(M, N, K)
    ws = _ws(nws, C) if nws > 0 else None
    call("gnf_gemm", ptr(A), a_strides[0], a_strides[1], ptr(Bm), ptr(Bmask), b_strides[0], b_strides[1], ptr(C),
         c_strides[0], c_strides[1], ptr(bias), ptr(Cmask), cm_strides[0], cm_strides[1], ptr(gate), g_strides[0],
         g_strides[1], 1 if relu else 0, M, N, K, ptr(ws), nws, stream())


class MLPFn(torch.autograd.Function):
    """y = L_n(relu(L_{n-1}(... relu(L_1(x))))), L_i(a) = a @ (mask_i * W_i)^T + b_i.

    One Function for every Linear/ReLU chain of the conditioners: CouplingMLP
    (CouplingConditioner.py:6-19), DAGMLP (DAGConditioner.py:7-20), MADE's masked linears
    (AutoregressiveConditioner.py:14-25, mask fused into the weight load instead of a
    mask*weight product per forward), MNISTCNN.fc1/fc2 (MLP.py:44-47).  The backward fuses the
    ReLU gate into the epilogue of the data-gradient kernel, the mask into the weight-gradient
    one and the bias gradient (column sums) into the same launch.  Every layer goes through the
    gnf_linear_* entry points: small batches run on weight-streaming kernels that evaluate a
    degree-structured mask (`degs[i] = (deg_out, deg_in, strict)`) instead of reading it."""

    @staticmethod
    def _spec(ctx, li):
        mk = ctx.masks[li] if ctx.masks is not None else None
        dg = ctx.degs[li] if ctx.degs is not None else None
        if dg is None:
            return ptr(mk), None, None, 0
        return ptr(mk), ptr(dg[0]), ptr(dg[1]), int(dg[2])

    @staticmethod
    def forward(ctx, x, masks, relu_in, degs, *params):
        """relu_in: x is itself the output of a ReLU (e.g. fc1 of the sparse embedding front): its gradient leaves this
        Function already gated by x > 0, fused into the data-gradient kernel's epilogue."""
        x = x.contiguous()
        n = len(params) // 2
        ctx.masks, ctx.degs, ctx.n, ctx.relu_in = masks, degs, n, bool(relu_in)
        acts = [x]
        a = x
        for li in range(n):
            W, b = params[2 * li].contiguous(), params[2 * li + 1]
            out_f, in_f = W.shape
            M = a.shape[0]
            y = _empty((M, out_f), x)
            nws = abi.load().gnf_linear_ws_bytes(M, out_f, in_f)
            ws = _ws(nws, x) if nws > 0 else None
            mk, do, di, st = MLPFn._spec(ctx, li)
            call("gnf_linear_fwd", ptr(a), ptr(W), ptr(b), mk, do, di, st, 1 if li < n - 1 else 0, ptr(y), M, out_f, in_f,
                 ptr(ws), nws, stream())
            a = y
            if li < n - 1:
                acts.append(y)
        ctx.save_for_backward(*acts, *params)
        return a

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n
        saved = ctx.saved_tensors
        acts, params = saved[:n], saved[n:]
        g = gy.contiguous()
        grads = [None] * (2 * n)
        gx = None
        gb_ready = None                     # bias gradient of layer li, already produced by layer li + 1's launch
        for li in range(n - 1, -1, -1):
            W = params[2 * li].contiguous()
            out_f, in_f = W.shape
            a = acts[li]
            M = a.shape[0]
            nws = abi.load().gnf_linear_ws_bytes(M, out_f, in_f)
            ws = _ws(nws, W) if nws > 0 else None
            mk, do, di, st = MLPFn._spec(ctx, li)
            want_w, want_b = ctx.needs_input_grad[4 + 2 * li], ctx.needs_input_grad[5 + 2 * li]
            want_x = li > 0 or ctx.needs_input_grad[0]
            gW = grad_out(W) if want_w else None
            gb = gb_ready
            if gb is None and want_w and want_b:
                gb = grad_out(params[2 * li + 1])
            ga = _empty((M, in_f), W) if want_x else None
            gate = a if (li > 0 or ctx.relu_in) else None
            gb_next = None
            if want_w and want_x:
                # the column sums of this layer's data gradient ARE the bias gradient of the layer below
                if (li > 0 and ctx.needs_input_grad[5 + 2 * (li - 1)] and ctx.needs_input_grad[4 + 2 * (li - 1)]
                        and abi.load().gnf_linear_gxsum_fused(M, out_f, in_f, int(mk is not None or do is not None))):
                    gb_next = grad_out(params[2 * li - 1])
                call("gnf_linear_bwd", ptr(g), ptr(W), ptr(a), mk, do, di, st, ptr(gate), ptr(ga), ptr(gW),
                     None if gb_ready is not None else ptr(gb), ptr(gb_next), M, out_f, in_f, ptr(ws), nws, stream())
            elif want_w:
                call("gnf_linear_bwd_w", ptr(g), ptr(a), mk, do, di, st, ptr(gW), None if gb_ready is not None else ptr(gb), M,
                     out_f, in_f, ptr(ws), nws, stream())
            elif want_x:
                call("gnf_linear_bwd_x", ptr(g), ptr(W), mk, do, di, st, ptr(gate), ptr(ga), M, out_f, in_f, ptr(ws), nws,
                     stream())
            grads[2 * li], grads[2 * li + 1] = gW, gb
            if want_b and not want_w:
                grads[2 * li + 1] = colsum(g)
            gb_ready = gb_next
            if want_x:
                g = ga
                if li == 0:
                    gx = ga
        return (gx, None, None, None, *grads)


def mlp(x, layers, masks=None, relu_in=False, degs=None):
    """layers: list of (weight, bias) parameter pairs; masks: per-layer [out, in] 0/1 tensors (or None); degs: per-layer
    (deg_out [out], deg_in [in], strict) with mask[o][i] = deg_in[i] <= deg_out[o] (strict: <) where the caller has
    verified that the mask tensor has that structure, else None."""
    flat = [p for Wb in layers for p in Wb]
    return MLPFn.apply(x, masks, relu_in, degs, *flat)


# ----------------------------------------------------------------------------- MNISTCNN conv front
class MnistConvFn(torch.autograd.Function):
    """flatten(max_pool2d(conv2(relu(conv1(e))), 2)) for 28x28 single-channel images
    (models/MLP.py:36-43), fused in LDS; backward recomputes conv1 in-kernel."""

    @staticmethod
    def forward(ctx, e, W1, b1, W2, b2, exact_ties=False):
        """exact_ties: direct-convolution forward whose pool argmax follows torch's first-maximum rule on exactly tied
        windows (include/gnf_hip.h); the Winograd forward otherwise"""
        e = e.contiguous()
        n = e.shape[0]
        W1c, b1c, W2c, b2c = W1.contiguous(), b1.contiguous(), W2.contiguous(), b2.contiguous()
        pooled = _empty((n, 2304), e)
        arg = torch.empty((n, 2304), dtype=torch.uint8, device=e.device)
        call("gnf_mnistcnn_conv_fwd", ptr(e), ptr(W1c), ptr(b1c), ptr(W2c), ptr(b2c), ptr(pooled), abi.rawptr(arg), n,
             int(bool(exact_ties)), stream())
        ctx.save_for_backward(e, W1c, b1c, W2c, b2c, arg)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        e, W1, b1, W2, b2, arg = ctx.saved_tensors
        n = e.shape[0]
        gp = gp.contiguous()
        ge = _empty((n, 784), e)
        gW1, gb1, gW2, gb2 = grad_out(W1), grad_out(b1), grad_out(W2), grad_out(b2)
        nws = abi.load().gnf_mnistcnn_conv_bwd_ws_bytes(n)
        ws = _ws(nws, e)
        call("gnf_mnistcnn_conv_bwd", ptr(e), ptr(W1), ptr(b1), ptr(W2), ptr(gp), abi.rawptr(arg), ptr(ge), ptr(gW1),
             ptr(gb1), ptr(gW2), ptr(gb2), abi.rawptr(ws), nws, n, stream())
        return ge, gW1, gb1, gW2, gb2, None


def crop_origin(p):
    """cell origin of the 5x5 pooled block that can deviate from the background for a pixel row / column p"""
    return min(max((p - 6) >> 1, 0), 7)


def mnist_window_mask(device, kernel=2):
    """[784,784] bool: True where pixel j lies in the (2 kernel + 1)^2 window around pixel i (the support
    MNIST_A_prior(28, kernel) allows, diagonal included)"""
    r = torch.arange(28, device=device).repeat_interleave(28)
    c = torch.arange(28, device=device).repeat(28)
    return ((r[:, None] - r[None, :]).abs() <= kernel) & ((c[:, None] - c[None, :]).abs() <= kernel)


class SparseRows:
    """Masked copies to evaluate, in the order the sparse kernels want them: sorted by crop origin, with the
    (first output row, row count) table of the 64 origins for a batch of B samples."""

    def __init__(self, rows, B, device):
        rows = [int(r) for r in rows]
        origin = [8 * crop_origin(r // 28) + crop_origin(r % 28) for r in rows]
        order = sorted(range(len(rows)), key=lambda k: (origin[k], rows[k]))
        self.R, self.B = len(rows), B
        self.identity = order == list(range(len(rows)))         # the caller's rows already are in the kernels' order
        self.pix = torch.tensor([rows[k] for k in order], dtype=torch.int32, device=device)
        inv = [0] * len(rows)
        for pos, k in enumerate(order):
            inv[k] = pos
        self.unsort = torch.tensor(inv, dtype=torch.long, device=device)   # position of rows[k] in the sorted order
        self.order = torch.tensor(order, dtype=torch.long, device=device)  # caller's index of the copy at a sorted position
        counts = [0] * 64
        for g in origin:
            counts[g] += 1
        table, first = [], 0
        for g in range(64):
            table += [first * B, counts[g] * B]
            first += counts[g]
        self.groups = torch.tensor(table, dtype=torch.int32, device=device)
        self.max_group_rows = max(max(counts) * B, 1)
        # chunks of <= 512 output rows inside each origin's range (the backward's per-chunk fc1 weight gradients)
        chunks, per_origin = [], []
        for g in range(64):
            first, rows_g = table[2 * g], table[2 * g + 1]
            per_origin += [len(chunks) // 2, (rows_g + 511) // 512]
            for o in range(0, rows_g, 512):
                chunks += [first + o, min(512, rows_g - o)]
        self.n_kgroups = len(chunks) // 2
        self.kgroups = torch.tensor(chunks or [0, 0], dtype=torch.int32, device=device)
        self.origin_chunks = torch.tensor(per_origin, dtype=torch.int32, device=device)


SPARSE_REUSE_TABLES = True     # the sparse front's backward reads the tables its forward built (False: builds them again; tests)


class MnistSparseFn(torch.autograd.Function):
    """relu(fc1(flatten(maxpool(conv2(relu(conv1(x * P[i]))))))) for the masked copies i of a SparseRows, P zero outside
    the 5x5 pixel windows: [R*B, F] in the SparseRows' sorted order (row = position * B + sample).  Differentiable w.r.t.
    the six network parameters only (x and P are treated as constants: frozen deterministic gate)."""

    @staticmethod
    def forward(ctx, x, P, sr, pre_gated, W1, b1, W2, b2, Wfc1, bfc1, prep=None, grad_mode=True):
        """pre_gated: the consumer returns the cotangent of h1 already multiplied by [h1 > 0] (MLPFn's relu_in);
        prep: the parameter-only tables of mnistcnn_sparse_prepare (inference: they are not rebuilt per call)"""
        x, P = x.contiguous(), P.contiguous()
        ws_ = [t.contiguous() for t in (W1, b1, W2, b2, Wfc1, bfc1)]
        F = Wfc1.shape[0]
        n = sr.R * sr.B
        h1 = _empty((n, F), x)
        # (needs_input_grad reports the parameters' requires_grad even under torch.no_grad(), and inside forward() the
        # grad mode is always off: the caller's mode comes in as an argument -- without it every sampling / evaluation pass
        # saved the pooled blocks and the argmax bytes of a backward that never comes)
        train = bool(grad_mode) and any(ctx.needs_input_grad[4:10])
        if prep is not None and not train:
            nws = n * 400 * 4
            ws = _ws(nws, x)
            call("gnf_mnistcnn_sparse_fwd_prepared", ptr(x), sr.B, ptr(P), abi.rawptr(sr.pix), sr.R, abi.rawptr(sr.groups),
                 sr.max_group_rows, *[ptr(t) for t in ws_[:4]], F, abi.rawptr(prep), ptr(h1), abi.rawptr(ws), nws, stream())
            return h1
        ctx.pre_gated = bool(pre_gated)
        if not train:
            nws = abi.load().gnf_mnistcnn_sparse_ws_bytes(n, F)
            ws = _ws(nws, x)
            call("gnf_mnistcnn_sparse_fwd", ptr(x), sr.B, ptr(P), abi.rawptr(sr.pix), sr.R, abi.rawptr(sr.groups),
                 sr.max_group_rows, *[ptr(t) for t in ws_], F, ptr(h1), None, None, abi.rawptr(ws), nws, stream())
            return h1
        pd = _empty((n, 400), x)
        arg = torch.empty((n, 400), dtype=torch.uint8, device=x.device)
        # training: ONLY the parameter-only tables (fc1 column blocks per crop origin, background response: 13 MB) are this
        # call's own and stay alive for the backward, which reads them instead of building them again; the pooled blocks go
        # straight to pd (round 5 held a whole workspace whose first n * 400 floats nobody touched: 125 MB per in-flight
        # forward at B = 100)
        nt = abi.load().gnf_mnistcnn_sparse_prep_bytes(F)
        tables = torch.empty(max(int(nt) // 4, 1), dtype=torch.float32, device=x.device)
        call("gnf_mnistcnn_sparse_fwd_train", ptr(x), sr.B, ptr(P), abi.rawptr(sr.pix), sr.R, abi.rawptr(sr.groups),
             sr.max_group_rows, *[ptr(t) for t in ws_], F, ptr(h1), ptr(pd), abi.rawptr(arg), abi.rawptr(tables), nt, stream())
        ctx.save_for_backward(x, P, *ws_[:5], pd, arg, h1)
        ctx.sr = sr
        ctx.tables = tables if SPARSE_REUSE_TABLES else None
        return h1

    @staticmethod
    def backward(ctx, gh1):
        x, P, W1, b1, W2, b2, Wfc1, pd, arg, h1 = ctx.saved_tensors
        sr = ctx.sr
        F = Wfc1.shape[0]
        n = sr.R * sr.B
        g = gh1.contiguous() if ctx.pre_gated else (gh1 * (h1 > 0)).contiguous()
        gW1, gb1, gW2, gb2 = grad_out(W1), grad_out(b1), grad_out(W2), grad_out(b2)
        gWf, gbf = grad_out(Wfc1), _empty((F,), x)
        nws = abi.load().gnf_mnistcnn_sparse_bwd_ws_bytes(n, F, sr.n_kgroups)
        ws = _ws(nws, x)
        tables = getattr(ctx, "tables", None)
        call("gnf_mnistcnn_sparse_bwd_tables", ptr(x), sr.B, ptr(P), abi.rawptr(sr.pix), sr.R, abi.rawptr(sr.groups),
             sr.max_group_rows, abi.rawptr(sr.kgroups), sr.n_kgroups, abi.rawptr(sr.origin_chunks), ptr(W1), ptr(b1), ptr(W2),
             ptr(b2), ptr(Wfc1), F, abi.rawptr(tables) if tables is not None else None, ptr(pd), abi.rawptr(arg), ptr(g),
             ptr(gW1), ptr(gb1), ptr(gW2), ptr(gb2), ptr(gWf), ptr(gbf), abi.rawptr(ws), nws, stream())
        ctx.tables = None
        return None, None, None, None, gW1, gb1, gW2, gb2, gWf, gbf, None, None


def mnistcnn_sparse_fwd(x, P, sr, W1, b1, W2, b2, Wfc1, bfc1, pre_gated=False, prep=None):
    return MnistSparseFn.apply(x, P, sr, pre_gated, W1, b1, W2, b2, Wfc1, bfc1, prep, torch.is_grad_enabled())


def sparse_fc12_fits(Wfc1, Wfc2):
    """the widths the one-launch fc1 + ReLU + fc2 kernel is built for (the reference's MNISTCNN: 2304 -> 128 -> <= 32);
    GNF_SPARSE_FC12=0 keeps the grouped GEMM + tall-layer pair (A/B runs)"""
    import os
    return (Wfc1.shape[0] == 128 and Wfc2.shape[1] == 128 and 1 <= Wfc2.shape[0] <= 32
            and os.environ.get("GNF_SPARSE_FC12", "1") != "0")


def mnistcnn_sparse_fwd_fc2(x, P, sr, W1, b1, W2, b2, prep, Wfc2, bfc2):
    """fc2(relu(fc1(conv front))) of the masked copies of `sr` against prepared tables: [R*B, out_d], inference only
    (MLP.py:36-47 on the deterministic-gate copies, crop kernel + ONE launch for both linear layers)"""
    x, P = x.detach().contiguous(), P.detach().contiguous()
    ts = [t.detach().contiguous() for t in (W1, b1, W2, b2, Wfc2, bfc2)]
    n, out_d = sr.R * sr.B, Wfc2.shape[0]
    h2 = _empty((n, out_d), x)
    nws = max(n, 1) * 400 * 4
    ws = _ws(nws, x)
    call("gnf_mnistcnn_sparse_fwd_prepared_fc2", ptr(x), sr.B, ptr(P), abi.rawptr(sr.pix), sr.R, abi.rawptr(sr.groups),
         sr.max_group_rows, *[ptr(t) for t in ts[:4]], 128, abi.rawptr(prep), ptr(ts[4]), ptr(ts[5]), out_d, ptr(h2),
         abi.rawptr(ws), nws, stream())
    return h2


def mnistcnn_sparse_prepare(b1, W2, b2, Wfc1, bfc1):
    """the parameter-only tables of the sparse front (fc1 weight columns per crop origin, background responses), built
    once for a caller that evaluates the front many times with unchanged parameters (the levels of a sampling pass)"""
    ts = [t.detach().contiguous() for t in (b1, W2, b2, Wfc1, bfc1)]
    F = Wfc1.shape[0]
    nb = abi.load().gnf_mnistcnn_sparse_prep_bytes(F)
    prep = _ws(nb, Wfc1)
    call("gnf_mnistcnn_sparse_prepare", *[ptr(t) for t in ts], F, abi.rawptr(prep), nb, stream())
    return prep


class PermuteRowsFn(torch.autograd.Function):
    """y = x[perm] along dim 0 with a gather in both directions (inv = inverse permutation): autograd's own backward
    of an index is a sort + scatter-add"""

    @staticmethod
    def forward(ctx, x, perm, inv):
        ctx.save_for_backward(inv)
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        return g.index_select(0, inv), None, None


# ----------------------------------------------------------------------------- DAG gate
IMP_RAW, IMP_SOFT, IMP_HARD_SOFT, IMP_HARD_SQ = 0, 1, 2, 3
GATE_DET, GATE_GUMBEL, GATE_NOISE = 0, 1, 2


class DagGateFn(torch.autograd.Function):
    """e[b*d+i, :] = x[b, :] * gate(importance(A[i, :])) (+ one-hot(i))
    (models/Conditionners/DAGConditioner.py:94-166)."""

    @staticmethod
    def forward(ctx, x, A, imp_mode, gate_mode, h_thresh, temperature, hot, u1, u2, seed, offset):
        x = x.contiguous()
        A = A.contiguous()
        B, d = x.shape
        ld = 2 * d if hot else d
        e = _empty((B * d, ld), x)
        u1 = u1.contiguous() if u1 is not None else None
        u2 = u2.contiguous() if u2 is not None else None
        nws = abi.load().gnf_dag_gate_fwd_ws_bytes(d)
        keep = A.requires_grad or x.requires_grad          # a backward will follow: its own buffer keeps the (i, j) table
        ws = torch.empty(max(int(nws) // 4, 1), dtype=torch.float32, device=x.device) if keep else _ws(nws, x)
        call("gnf_dag_gate_fwd", ptr(x), ptr(A), ptr(e), ld, imp_mode, gate_mode, float(h_thresh), float(temperature),
             ptr(u1), ptr(u2), seed, offset, int(hot), ptr(ws), B, d, stream())
        ctx.tab = ws if (keep and B > 0) else None
        ctx.save_for_backward(x, A, u1, u2)
        ctx.cfg = (imp_mode, gate_mode, float(h_thresh), float(temperature), ld, seed, offset)
        return e

    @staticmethod
    def backward(ctx, ge):
        x, A, u1, u2 = ctx.saved_tensors
        imp_mode, gate_mode, h_thresh, temperature, ld, seed, offset = ctx.cfg
        B, d = x.shape
        ge = ge.contiguous()
        gA, finish, _ = grad_out_shared(A) if ctx.needs_input_grad[1] else (None, None, False)
        gx = _empty((B, d), x) if ctx.needs_input_grad[0] else None
        ws = _ws(abi.load().gnf_dag_gate_bwd_ws_bytes(B, d), x)
        call("gnf_dag_gate_bwd", ptr(x), ptr(A), ptr(ge), ld, imp_mode, gate_mode, h_thresh, temperature, ptr(u1),
             ptr(u2), seed, offset, ptr(ctx.tab), ptr(gA), ptr(gx), ptr(ws), B, d, stream())
        return gx, (finish(gA) if gA is not None else None), None, None, None, None, None, None, None, None, None


class DagConvFrontFn(torch.autograd.Function):
    """flatten(max_pool2d(conv2(relu(conv1(e))), 2)) of the masked copies e[b*d+i, :] = x[b, :] * gate(importance(A[i, :]))
    -- DagGateFn and MnistConvFn as ONE autograd node (DAGConditioner.py:94-166,169 -> MLP.py:36-43), which is what lets the
    backward drop the structural zeros: dL/dA[i,j] = dP/dA[i,j] * (...) and dP/dA is exactly 0 wherever A is (97.2 % of
    MNIST_A_prior, DAGConditioner.py:118-119), so when x wants no gradient the conv backward writes the cotangent of e at
    the <= 32 columns per row the forward's plan lists (10 MB instead of 243 MB at B = 100, and it skips the W1^T dpre1
    products nobody reads) and the gate backward visits those (i, j) pairs only.  The plan is rebuilt from A on the device
    in every forward; a row with more columns than the plan holds sends both kernels down their dense code (decided on the
    device); x.requires_grad takes the dense entry points."""

    @staticmethod
    def forward(ctx, x, A, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset, W1, b1, W2, b2, exact_ties,
                grad_mode):
        x, A = x.contiguous(), A.contiguous()
        B, d = x.shape
        if d != 784:
            raise abi.GnfError("the fused masked-image front is the 28 x 28 MNISTCNN one: d = %d" % d)
        n = B * d
        e = _empty((n, d), x)
        u1 = u1.contiguous() if u1 is not None else None
        u2 = u2.contiguous() if u2 is not None else None
        lib = abi.load()
        tab = torch.empty(max(int(lib.gnf_dag_gate_fwd_ws_bytes(d)) // 4, 1), dtype=torch.float32, device=x.device)
        want_plan = bool(grad_mode) and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0] and B > 0
        plan, nplan = None, 0
        if want_plan:
            nplan = int(lib.gnf_dag_gate_plan_bytes(d))
            plan = torch.empty(nplan // 4, dtype=torch.int32, device=x.device)
        call("gnf_dag_gate_fwd_plan", ptr(x), ptr(A), ptr(e), d, imp_mode, gate_mode, float(h_thresh), float(temperature),
             ptr(u1), ptr(u2), seed, offset, 0, ptr(tab), abi.rawptr(plan) if plan is not None else None, nplan, B, d,
             stream())
        W1c, b1c, W2c, b2c = W1.contiguous(), b1.contiguous(), W2.contiguous(), b2.contiguous()
        pooled = _empty((n, 2304), x)
        arg = torch.empty((n, 2304), dtype=torch.uint8, device=x.device)
        call("gnf_mnistcnn_conv_fwd", ptr(e), ptr(W1c), ptr(b1c), ptr(W2c), ptr(b2c), ptr(pooled), abi.rawptr(arg), n,
             int(bool(exact_ties)), stream())
        ctx.save_for_backward(x, A, u1, u2, e, W1c, b1c, W2c, b2c, arg)
        ctx.tab, ctx.plan = (tab if B > 0 else None), plan
        ctx.cfg = (imp_mode, gate_mode, float(h_thresh), float(temperature), seed, offset)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        x, A, u1, u2, e, W1, b1, W2, b2, arg = ctx.saved_tensors
        imp_mode, gate_mode, h_thresh, temperature, seed, offset = ctx.cfg
        B, d = x.shape
        n = B * d
        lib = abi.load()
        gp = gp.contiguous()
        ge = _empty((n, d), x)                 # (with a plan: touched only if one of its rows overflows)
        gW1, gb1, gW2, gb2 = grad_out(W1), grad_out(b1), grad_out(W2), grad_out(b2)
        nws = lib.gnf_mnistcnn_conv_bwd_ws_bytes(n)
        ws = _ws(nws, x)
        plan = ctx.plan
        gec = _empty((n, abi.DAG_PLAN_KC), x) if plan is not None else None
        call("gnf_mnistcnn_conv_bwd_cols", ptr(e), ptr(W1), ptr(b1), ptr(W2), ptr(gp), abi.rawptr(arg), ptr(ge),
             abi.rawptr(plan) if plan is not None else None, d, ptr(gec), ptr(gW1), ptr(gb1), ptr(gW2), ptr(gb2),
             abi.rawptr(ws), nws, n, stream())
        gA, finish, acc = (grad_out_shared(A, accumulate_ok=plan is not None) if ctx.needs_input_grad[1]
                           else (None, None, False))
        gx = _empty((B, d), x) if ctx.needs_input_grad[0] else None
        if plan is not None:
            ws2 = _ws(lib.gnf_dag_gate_bwd_cols_ws_bytes(B, d), x)
            call("gnf_dag_gate_bwd_cols", ptr(x), ptr(ge), ptr(gec), abi.rawptr(plan), imp_mode, gate_mode, temperature,
                 ptr(u1), ptr(u2), seed, offset, ptr(ctx.tab), ptr(gA), int(acc), ptr(ws2), B, d, stream())
        elif gA is not None or gx is not None:
            ws2 = _ws(lib.gnf_dag_gate_bwd_ws_bytes(B, d), x)
            call("gnf_dag_gate_bwd", ptr(x), ptr(A), ptr(ge), d, imp_mode, gate_mode, h_thresh, temperature, ptr(u1),
                 ptr(u2), seed, offset, ptr(ctx.tab), ptr(gA), ptr(gx), ptr(ws2), B, d, stream())
        return (gx, (finish(gA) if gA is not None else None), None, None, None, None, None, None, None, None,
                gW1, gb1, gW2, gb2, None, None)


def dag_conv_front(x, A, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset, W1, b1, W2, b2,
                   exact_ties=False):
    return DagConvFrontFn.apply(x, A, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset, W1, b1, W2, b2,
                                exact_ties, torch.is_grad_enabled())


# ----------------------------------------------------------------------------- Monotonic (UMNN) normalizer
_CC = {}


def cc_rule(nb_steps, device):
    """Clenshaw-Curtis weights / nodes, built on the host in fp64 with the construction of
    UMNN 1.0's compute_cc_weights (third-party, absent from the reference tree; restated
    from its published algorithm -- parity unpinned), cast to fp32."""
    key = (int(nb_steps), str(device))
    if key not in _CC:
        S = int(nb_steps)
        k = np.arange(0, S + 1, dtype=np.float64).reshape(-1, 1)
        lam = np.cos((k @ k.T) * math.pi / S)
        lam[:, 0] = .5
        lam[:, -1] = .5 * lam[:, -1]
        lam = lam * 2 / S
        W = k.copy()
        odd = np.arange(1, S + 1, 2)
        W[odd] = 0
        W = 2 / (1 - W ** 2)
        W[0] = 1
        W[odd] = 0
        w = (lam.T @ W).reshape(-1)
        t = np.cos(np.arange(0, S + 1) * math.pi / S)
        _CC[key] = (torch.tensor(w, dtype=torch.float32, device=device),
                    torch.tensor(t, dtype=torch.float32, device=device))
    return _CC[key]


def _mono_net(params):
    """params: [W0, b0, W1, b1, ...] contiguous fp32 HIP tensors -> gnf_mono_net."""
    nl = len(params) // 2
    if nl < 2 or nl > abi.MONO_MAX_LAYERS:
        raise abi.GnfError("integrand net needs 2..%d Linear layers" % abi.MONO_MAX_LAYERS)
    net = abi.MonoNet()
    net.nl = nl
    net.dims[0] = params[0].shape[1]
    for l in range(nl):
        W, b = params[2 * l], params[2 * l + 1]
        if W.shape[1] != net.dims[l]:
            raise abi.GnfError("integrand net layer %d: in_features mismatch" % l)
        net.dims[l + 1] = W.shape[0]
        net.W[l] = ptr(W).value
        net.b[l] = ptr(b).value
    return net


def _mono_pack(net, like):
    """the kernels' image of the integrand net's parameters (padded, transposed and fragment-major copies)"""
    nfl = abi.load().gnf_monotonic_pack_floats(ctypes.byref(net))
    if nfl < 0:
        abi.check(int(nfl), "gnf_monotonic_pack_floats")
    pack = _empty((int(nfl),), like)
    call("gnf_monotonic_pack", ctypes.byref(net), ptr(pack), stream())
    return pack


def monotonic_pack(params, like):
    """the weight image for `monotonic_inverse(..., pack=...)`: a caller that inverts many times with unchanged parameters
    (the 109 DAG levels of one MNIST sampling pass) packs once.  (Not cached behind the caller's back: the fused Adam
    launch and a replayed hipGraph rewrite the parameters without touching torch's version counters.)"""
    params = [p.detach().contiguous() for p in params]
    return _mono_pack(_mono_net(params), like)


class MonotonicFn(torch.autograd.Function):
    """z = int_0^x f(t;h) dt + h[...,0],  jac = f(x;h)
    (models/Normalizers/MonotonicNormalizer.py:51-66 + UMNN NeuralIntegral, incl. its
    Leibniz-rule x-gradient)."""

    @staticmethod
    def forward(ctx, x, h, nb_steps, *params):
        x = x.contiguous()
        params = [p.contiguous() for p in params]
        B, d = x.shape
        net = _mono_net(params)
        if h.shape[2] != net.dims[0] - 1:
            raise abi.GnfError("cond_size %d != integrand net input %d - 1" % (h.shape[2], net.dims[0]))
        pack = _mono_pack(net, x)
        w, t = cc_rule(nb_steps, x.device)
        z, jac = _empty((B, d), x), _empty((B, d), x)
        call("gnf_monotonic_fwd", ptr(pack), ctypes.byref(net), ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2),
             ptr(w), ptr(t), int(nb_steps), ptr(z), ptr(jac), B, d, stream())
        ctx.save_for_backward(x, h, pack, *params)
        ctx.S = int(nb_steps)
        return z, jac

    @staticmethod
    def backward(ctx, gz, gjac):
        x, h, pack, *params = ctx.saved_tensors
        B, d = x.shape
        S = ctx.S
        net = _mono_net(params)
        if h.stride(0) != d * h.stride(1):
            h = h.contiguous()          # element stride must collapse for the d W1 GEMM
        w, t = cc_rule(S, x.device)
        gz = gz.contiguous() if gz is not None else torch.zeros_like(x)
        gjac = gjac.contiguous() if gjac is not None else None
        gx = _empty((B, d), x)
        gh = _empty(tuple(h.shape), x)
        gparams = [grad_out(p) for p in params]
        nl = net.nl
        gW = (ctypes.c_void_p * nl)(*[gparams[2 * l].data_ptr() for l in range(nl)])
        gb = (ctypes.c_void_p * nl)(*[gparams[2 * l + 1].data_ptr() for l in range(nl)])
        nbytes = abi.load().gnf_monotonic_bwd_ws_bytes(ctypes.byref(net), S, B, d)
        if nbytes < 0:
            abi.check(int(nbytes), "gnf_monotonic_bwd_ws_bytes")
        ws = _ws(nbytes, x)
        call("gnf_monotonic_bwd", ptr(pack), ctypes.byref(net), ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2),
             ptr(w), ptr(t), S, ptr(gz), ptr(gjac), ptr(gx), ptr(gh), gh.stride(0), gh.stride(1), gh.stride(2),
             gW, gb, ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, B, d, stream())
        return (gx, gh, None, *gparams)


def monotonic_inverse(z, h, nb_steps, params, pack=None, out=None, out_cols=None):
    """20-step bisection on [-20, 20], the quadrature fused in the kernel
    (MonotonicNormalizer.py:69-83).  out / out_cols: write the TRANSPOSED result into columns out_cols of `out` (the
    level loop of NormalizingFlowStep.invert: z is [level rows, batch], the sample is [batch, d])."""
    z = z.contiguous()
    params = [p.detach().contiguous() for p in params]
    B, d = z.shape
    net = _mono_net(params)
    if pack is None:
        pack = _mono_pack(net, z)
    w, t = cc_rule(nb_steps, z.device)
    if out is not None:
        # element (b, j) of this [B, d] problem -> out[j, out_cols[b]]  (out [d, width] contiguous, out_cols int32 [B])
        assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == d and out_cols.dtype == torch.int32
        assert out_cols.numel() == B and out_cols.device == z.device
        call("gnf_monotonic_inv_scatter", ptr(pack), ctypes.byref(net), ptr(z), ptr(h), h.stride(0), h.stride(1),
             h.stride(2), ptr(w), ptr(t), int(nb_steps), ptr(out), abi.rawptr(out_cols), out.shape[1], B, d, stream())
        return out
    x = _empty((B, d), z)
    call("gnf_monotonic_inv", ptr(pack), ctypes.byref(net), ptr(z), ptr(h), h.stride(0), h.stride(1), h.stride(2),
         ptr(w), ptr(t), int(nb_steps), ptr(x), B, d, stream())
    return x


class ModuleIntegralFn(torch.autograd.Function):
    """z = int_0^x f(t; h) dt for a USER-SUPPLIED integrand module (MonotonicNormalizer(integrand_net=<nn.Module>),
    reference MonotonicNormalizer.py:44-48,51-63): the module is honoured as it is and evaluated through PyTorch-ROCm
    on the device, at all Clenshaw-Curtis nodes in one batch ("CCParallel" form; "CC" gives the same numbers up to
    summation order).  Only the reference's own IntegrandNet architecture is fused into the gfx950 kernel.

    Same conventions as the fused kernel / UMNN's NeuralIntegral (restated, parity unpinned): forward without a graph;
    backward re-evaluates the integrand at the nodes to get d/d(theta) and d/dh by the same quadrature weighted by
    grad * x / 2, and uses the Leibniz rule for the upper limit, dz/dx = f(x; h)."""

    @staticmethod
    def forward(ctx, x, hflat, module, nb_steps, *params):
        w, t = cc_rule(nb_steps, x.device)
        S1 = w.numel()
        B, d = x.shape
        with torch.no_grad():
            nodes = (x.unsqueeze(0) * ((t.view(S1, 1, 1) + 1.) * .5)).reshape(S1 * B, d)
            f = module(nodes, hflat.unsqueeze(0).expand(S1, B, -1).reshape(S1 * B, -1)).view(S1, B, d)
            z = (f * w.view(S1, 1, 1)).sum(0) * x * .5
        ctx.module, ctx.S = module, int(nb_steps)
        ctx.save_for_backward(x, hflat)
        return z

    @staticmethod
    def backward(ctx, gz):
        x, hflat = ctx.saved_tensors
        module = ctx.module
        w, t = cc_rule(ctx.S, x.device)
        S1 = w.numel()
        B, d = x.shape
        params = [p for p in module.parameters() if p.requires_grad]
        with torch.enable_grad():
            hr = hflat.detach().requires_grad_(True)
            nodes = (x.detach().unsqueeze(0) * ((t.view(S1, 1, 1) + 1.) * .5)).reshape(S1 * B, d)
            f = module(nodes, hr.unsqueeze(0).expand(S1, B, -1).reshape(S1 * B, -1)).view(S1, B, d)
            weight = (gz * x.detach() * .5).unsqueeze(0) * w.view(S1, 1, 1)
            grads = torch.autograd.grad((f * weight).sum(), [hr] + params, allow_unused=True)
            fx = module(x.detach(), hflat.detach())
        gx = gz * fx.detach() if ctx.needs_input_grad[0] else None
        return (gx, grads[0], None, None, *grads[1:])


def module_monotonic(x, h, module, nb_steps):
    """(z, jac) of MonotonicNormalizer.forward for a custom integrand module; h: [B, d, c]"""
    B, d = x.shape
    hflat = h.permute(0, 2, 1).contiguous().view(B, -1)          # cond-major, as the reference passes it (:55)
    params = [p for p in module.parameters() if p.requires_grad]
    z = ModuleIntegralFn.apply(x, hflat, module, int(nb_steps), *params) + h[:, :, 0]
    return z, module(x, hflat)


def module_monotonic_inverse(z, h, module, nb_steps):
    """the reference's bisection (MonotonicNormalizer.py:69-83): 20 halvings of [-20, 20], midpoint returned"""
    B, d = z.shape
    hflat = h.permute(0, 2, 1).contiguous().view(B, -1)
    lo = torch.full_like(z, -20.)
    hi = torch.full_like(z, 20.)
    h0 = h[:, :, 0]
    with torch.no_grad():
        for _ in range(20):
            mid = (lo + hi) * .5
            zm = ModuleIntegralFn.apply(mid, hflat, module, int(nb_steps)) + h0
            left = zm > z                                     # (:76-79) solution lies left of the midpoint
            hi = torch.where(left, mid, hi)
            lo = torch.where(left, lo, mid)
    return (lo + hi) * .5


# ----------------------------------------------------------------------------- Adam on a flat buffer
def adam_step(p, g, m, v, step, lr=1e-3, betas=(.9, .999), eps=1e-8, weight_decay=0., grad_scale=1.):
    call("gnf_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
         grad_scale, int(step), stream())


def adam_step_dev(p, g, m, v, step_dev, lr=1e-3, betas=(.9, .999), eps=1e-8, weight_decay=0., grad_scale=1.,
                  advance=True):
    """Adam with the step counter in device memory (int32 tensor, incremented by the call): graph-capturable."""
    if step_dev.dtype != torch.int32 or not step_dev.is_cuda or step_dev.numel() < 2 or not step_dev.is_contiguous():
        raise abi.GnfError("step_dev must be a contiguous int32 HIP tensor {steps taken, 0}")
    call("gnf_adam_step_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
         grad_scale, abi.rawptr(step_dev), int(bool(advance)), stream())


class PowerTraceFn(torch.autograd.Function):
    """tr(B^k) of the DAG acyclicity term (DAGConditioner.py:192-194) with the closed-form gradient
    d tr(B^k) / dB = k (B^(k-1))^T: one power B^(k-1) serves the value (tr(B^k) = sum_ij (B^(k-1))_ij B_ji) and the
    gradient, instead of autograd through every product of torch.matrix_power (6 instead of 18 d x d GEMMs per step
    at d = 784, k = 34).  The GEMMs themselves stay on the library."""

    @staticmethod
    def forward(ctx, B, k):
        k = int(k)
        ctx.k = k
        if k == 0:
            ctx.save_for_backward(None)
            return B.new_tensor(float(B.shape[0]))
        P = torch.matrix_power(B, k - 1)
        ctx.save_for_backward(P)
        return (P * B.t()).sum()

    @staticmethod
    def backward(ctx, g):
        (P,) = ctx.saved_tensors
        if ctx.k == 0:
            return None, None
        return (g * ctx.k) * P.t(), None



class DagLossFn(torch.autograd.Function):
    """dag_const (lambd h + c/2 h^2) + l1 mean|A| with h = tr((I + alpha A o A)^k) - d  (DAGConditioner.py:176-194,
    268-271): three fused launches around the library matrix power instead of ~40 elementwise ones; the conditioner's
    scalar buffers are read on the device (no host synchronisation).  Returns the loss; `.trace` of the ctx-less call is
    available through dag_loss_terms()."""

    @staticmethod
    def forward(ctx, A, alpha, alpha_factor, lambd, c, dag_const, l1_weight, k):
        A = A.contiguous()
        d = A.shape[0]
        k = int(k)
        f32 = lambda t: torch.as_tensor(t, dtype=torch.float32, device=A.device).detach().reshape(1)   # buffers or floats
        al, lm, cc, dc, l1 = f32(alpha), f32(lambd), f32(c), f32(dag_const), f32(l1_weight)
        Bm = _empty((d, d), A)
        call("gnf_dag_loss_prep", ptr(A), ptr(al), float(alpha_factor), ptr(Bm), d, stream())
        P = torch.matrix_power(Bm, k - 1) if k > 0 else None
        out4, ws = _empty((4,), A), _empty((512,), A)
        call("gnf_dag_loss_value", ptr(A), ptr(Bm), ptr(P), ptr(al), float(alpha_factor), ptr(lm), ptr(cc), ptr(dc),
             ptr(l1), k, ptr(out4), ptr(ws), d, stream())
        ctx.save_for_backward(A, P, out4)
        return out4[0]

    @staticmethod
    def backward(ctx, g):
        A, P, out4 = ctx.saved_tensors
        gA, finish, _ = grad_out_shared(A)
        call("gnf_dag_loss_bwd", ptr(A), ptr(P), ptr(out4), ptr(g.contiguous().reshape(1)), ptr(gA), A.shape[0], stream())
        return finish(gA), None, None, None, None, None, None, None
