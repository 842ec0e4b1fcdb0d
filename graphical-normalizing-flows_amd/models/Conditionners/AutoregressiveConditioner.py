import numpy as np
import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditioner import Conditioner, no_context


class MaskedLinear(nn.Linear):
    """nn.Linear with a fixed 0/1 `mask` buffer on the weights (reference
    AutoregressiveConditioner.py:14-25).  The product mask*weight is never materialised:
    the GEMM multiplies the mask in while it stages the weight tile."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self.register_buffer('mask', torch.ones(out_features, in_features))
        # the degrees the mask was built from (not part of the reference's state_dict): mask[o][i] = deg_in[i] <= deg_out[o]
        # (strict: <).  The small-batch kernels evaluate that rule instead of streaming the fp32 mask next to the weights.
        self.register_buffer('deg_out', torch.zeros(out_features), persistent=False)
        self.register_buffer('deg_in', torch.zeros(in_features), persistent=False)
        self.deg_strict = False
        self._deg_checked = None           # (mask version, deg versions) of the last verification, and its verdict
        self._deg_ok = False

    def set_mask(self, mask, deg_out=None, deg_in=None, strict=False):
        self.mask.data.copy_(torch.from_numpy(mask.astype(np.uint8).T))
        if deg_out is not None:
            self.deg_out.data.copy_(torch.as_tensor(np.asarray(deg_out), dtype=torch.float32))
            self.deg_in.data.copy_(torch.as_tensor(np.asarray(deg_in), dtype=torch.float32))
            self.deg_strict = bool(strict)
        self._deg_checked = None

    def degree_spec(self):
        """(deg_out, deg_in, strict) if the mask buffer IS the degree rule (checked once per change of the buffers -- a
        loaded checkpoint or a user's set_mask may carry any 0/1 pattern), else None: the kernels then read the mask."""
        key = (self.mask._version, self.deg_out._version, self.deg_in._version, self.mask.data_ptr(), self.deg_strict)
        if self._deg_checked != key:
            cmp = torch.lt if self.deg_strict else torch.le
            self._deg_ok = bool(torch.equal(self.mask, cmp(self.deg_in[None, :], self.deg_out[:, None]).to(self.mask.dtype)))
            self._deg_checked = key
        return (self.deg_out, self.deg_in, self.deg_strict) if self._deg_ok else None

    def forward(self, input):
        return ops.mlp(input, [(self.weight, self.bias)], [self.mask], degs=[self.degree_spec()])


def made_degrees(nin, hidden_sizes, random=False, natural_ordering=False, rng=None):
    """Degrees of the reference MADE (:79-87): the natural ordering with degrees nin - 1 - (i mod nin) (`random=False`, the
    only form the reference's AutoregressiveConditioner builds), or a sampled ordering / connectivity drawn from `rng` exactly
    as the reference draws it (input order: a permutation unless natural_ordering; hidden degrees uniform in
    [min of the layer below, nin - 1))."""
    if random:
        m = {-1: np.arange(nin) if natural_ordering else rng.permutation(nin)}
        for l, hsz in enumerate(hidden_sizes):
            m[l] = rng.randint(m[l - 1].min(), nin - 1, size=hsz)
        return m
    m = {-1: np.arange(nin)}
    for l, hsz in enumerate(hidden_sizes):
        m[l] = np.array([nin - 1 - (i % nin) for i in range(hsz)])
    return m


class MADE(nn.Module):
    """Masked autoencoder, reference AutoregressiveConditioner.py:28-109: natural ordering (`random=False`, what the
    conditioner builds) or sampled orderings (`random=True`), cycling through `num_masks` seeds on update_masks().  Every
    mask is a degree rule deg_in <= deg_out (< for the output layer), so the small-batch kernels evaluate it from the two
    degree vectors whatever the ordering; output neurons are chunked component-major."""

    def __init__(self, nin, hidden_sizes, nout, num_masks=1, natural_ordering=False, random=False, device="cpu"):
        super().__init__()
        self.random = random
        self.nin = nin
        self.nout = nout
        self.hidden_sizes = hidden_sizes
        assert self.nout % self.nin == 0, "nout must be integer multiple of nin"
        net = []
        hs = [nin] + hidden_sizes + [nout]
        for h0, h1 in zip(hs, hs[1:]):
            net.extend([MaskedLinear(h0, h1), nn.ReLU()])
        net.pop()
        self.net = nn.Sequential(*net)
        self.natural_ordering = natural_ordering
        self.num_masks = num_masks
        self.seed = 0                                    # cycles through the num_masks orderings (reference :68)
        self.m = {}
        self.update_masks()

    def update_masks(self):
        if self.m and self.num_masks == 1:
            return
        L = len(self.hidden_sizes)
        rng = np.random.RandomState(self.seed)           # (drawn from even when unused, as the reference does)
        self.seed = (self.seed + 1) % self.num_masks
        self.m = made_degrees(self.nin, self.hidden_sizes, self.random, self.natural_ordering, rng)
        masks = [self.m[l - 1][:, None] <= self.m[l][None, :] for l in range(L)]
        masks.append(self.m[L - 1][:, None] < self.m[-1][None, :])
        degs = [(self.m[l], self.m[l - 1], False) for l in range(L)]
        deg_last = self.m[-1]
        if self.nout > self.nin:
            masks[-1] = np.concatenate([masks[-1]] * int(self.nout / self.nin), axis=1)
            deg_last = np.concatenate([deg_last] * int(self.nout / self.nin))
        degs.append((deg_last, self.m[L - 1], True))
        for layer, mk, (do, di, strict) in zip(self.masked_layers(), masks, degs):
            layer.set_mask(mk, do, di, strict)
        # position of input dimension k in the sampled order (a public attribute of the reference's class, :99-101)
        self.i_map = self.m[-1].copy()
        for k in range(len(self.m[-1])):
            self.i_map[self.m[-1][k]] = k

    def masked_layers(self):
        return [l for l in self.net if isinstance(l, MaskedLinear)]

    def forward(self, x):
        ls = self.masked_layers()
        y = ops.mlp(x, [(l.weight, l.bias) for l in ls], [l.mask for l in ls], degs=[l.degree_spec() for l in ls])
        return y.view(x.shape[0], -1, x.shape[1]).permute(0, 2, 1)


class ConditionnalMADE(MADE):
    """(sic) reference AutoregressiveConditioner.py:115-141; cond_in must be 0."""

    def __init__(self, nin, cond_in, hidden_sizes, nout, num_masks=1, natural_ordering=False, random=False,
                 device="cpu"):
        super().__init__(nin + cond_in, hidden_sizes, nout, num_masks, natural_ordering, random, device)
        self.nin_non_cond = nin
        self.cond_in = cond_in

    def forward(self, x, context):
        no_context(context, self.cond_in)
        return super().forward(x)


class AutoregressiveConditioner(Conditioner):
    """reference AutoregressiveConditioner.py:144-154: h[b,i,:] depends on x[b,:i] only.
    Returns a permuted [B,d,hs] view (strides (hs*d, 1, d)) exactly like the reference; the
    normalizer kernels take the strides as they are."""

    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super(AutoregressiveConditioner, self).__init__()
        self.in_size = in_size
        self.masked_autoregressive_net = ConditionnalMADE(in_size, cond_in=cond_in, hidden_sizes=hidden,
                                                          nout=out_size * in_size)

    def forward(self, x, context=None):
        return self.masked_autoregressive_net(x, context)

    def depth(self):
        return self.in_size - 1
