"""Embedding networks of the reference (models/MLP.py) on the MI355X path.

Same class names, constructor arguments, attribute names and therefore `state_dict` keys (`net.K.*`, `conv1/conv2`,
`fc1/fc2[/fc3]`).  What differs is how they run: Linear/ReLU chains go through the fused MFMA GEMM chain
(gnf_hip.ops.mlp); the convolutional front of `MNISTCNN` on 28x28 single-channel images -- the embedding net of the
MNIST DAG flow, evaluated on B*d masked images per step -- is one LDS-resident Winograd/MFMA kernel pair
(gnf_hip.ops.MnistConvFn, csrc/gnf_mnistcnn.hip)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from gnf_hip import ops


def _linears(module):
    return [(m.weight, m.bias) for m in module if isinstance(m, nn.Linear)]


def _conv3x3(x, weight, bias):
    """valid 3x3 convolution of a whole batch as ONE im2col gather (strided `unfold` views + one copy) and one MFMA GEMM
    with the bias fused; same values as nn.Conv2d.  (F.unfold launches one im2col kernel per image: 49 000 launches per
    step on the 14x14 / 7x7 scales of the 3-scale MNIST flow.)"""
    n, c, h, w = x.shape
    cols = x.unfold(2, 3, 1).unfold(3, 3, 1)                                    # [n, c, h-2, w-2, 3, 3] view
    cols = cols.permute(0, 2, 3, 1, 4, 5).reshape(n * (h - 2) * (w - 2), c * 9)  # rows = output positions
    y = ops.mlp(cols, [(weight.flatten(1), bias)])                              # [n * positions, out]
    return y.view(n, h - 2, w - 2, weight.shape[0]).permute(0, 3, 1, 2)


class MLP(nn.Module):
    """in_d -> hidden... -> out_d with `act_f` between the Linear layers (reference :6-21)."""

    def __init__(self, in_d, hidden, out_d, act_f=nn.ReLU()):
        super().__init__()
        self.in_d, self.hiddens, self.out_d, self.act_f = in_d, hidden, out_d, act_f
        widths = [in_d] + list(hidden) + [out_d]
        mods = []
        for k in range(len(widths) - 1):
            mods.append(nn.Linear(widths[k], widths[k + 1]))
            if k + 2 < len(widths):
                mods.append(act_f)
        self.net = nn.Sequential(*mods)

    def forward(self, x, context=None):
        if isinstance(self.act_f, nn.ReLU):
            return ops.mlp(x, _linears(self.net))
        return self.net(x)                       # other activations: plain torch


class MNISTCNN(nn.Module):
    """conv3x3(C->16) ReLU conv3x3(16->16) maxpool2 flatten fc(fc_l[0]->fc_l[1]) ReLU fc(->out_d)  (reference :24-48;
    its two Dropout2d members are constructed but never applied there, likewise here)."""

    def __init__(self, out_d=10, fc_l=[2304, 128], size_img=[1, 28, 28]):
        super().__init__()
        self.size_img, self.out_d = size_img, out_d
        self.conv1 = nn.Conv2d(size_img[0], 16, 3, 1)
        self.conv2 = nn.Conv2d(16, 16, 3, 1)
        self.dropout1, self.dropout2 = nn.Dropout2d(0.25), nn.Dropout2d(0.5)
        self.fc1 = nn.Linear(fc_l[0], fc_l[1])
        self.fc2 = nn.Linear(fc_l[1], out_d)
        # max-pool windows whose four values are exactly equal (constant image regions) follow torch's first-maximum rule
        # only on the direct-convolution forward; DAGConditioner sets this for deterministic gates (their masked copies
        # are exactly zero outside the allowed inputs), the stochastic gate's copies have no exact ties
        self.exact_pool_ties = False

    def _fused_front(self, x):
        """the Winograd/MFMA kernels cover exactly the 1 x 28 x 28 -> 16 x 12 x 12 case on the GPU"""
        return (x.is_cuda and list(self.size_img) == [1, 28, 28] and x.shape[-1] == 784
                and self.conv1.weight.shape == (16, 1, 3, 3) and self.conv2.weight.shape == (16, 16, 3, 3))

    def _embeddable(self, x):
        """single-channel images of 6..28 pixels a side with the 1->16->16 3x3 convolutions: run on the 28x28 kernels"""
        c, h, w = self.size_img
        return (x.is_cuda and c == 1 and 6 <= h <= 28 and 6 <= w <= 28 and x.shape[-1] == h * w
                and self.conv1.weight.shape == (16, 1, 3, 3) and self.conv2.weight.shape == (16, 16, 3, 3))

    def supports_sparse(self, x):
        """the sparse masked-copy front (gnf_hip.ops.mnistcnn_sparse_fwd) covers the 28x28 net with fc1 on 2304 inputs"""
        return (self._fused_front(x) and self.fc1.in_features == 2304 and self.fc1.out_features % 4 == 0)

    def sparse_rows(self, x, P, sr, variable_major=False):
        """Embeddings [B, R, out_d] of the masked copies x * P[i], i in the gnf_hip.ops.SparseRows `sr` (returned in the
        caller's order), for an importance matrix P that is zero outside the 5x5 pixel windows: only the 14x14 crop
        that can differ from the all-zero image is convolved.  Differentiable w.r.t. the network's
        parameters (not x, not P)."""
        if (not torch.is_grad_enabled() and self._held_prep is not None
                and ops.sparse_fc12_fits(self.fc1.weight, self.fc2.weight)):
            # inference against held tables: crop kernel + one launch for fc1 + ReLU + fc2
            out = ops.mnistcnn_sparse_fwd_fc2(x.view(-1, 784), P, sr, self.conv1.weight, self.conv1.bias, self.conv2.weight,
                                              self.conv2.bias, self._held_prep, self.fc2.weight, self.fc2.bias)
            out = out.view(sr.R, sr.B, -1)
            if not sr.identity:
                out = out.index_select(0, sr.unsort)
            return out if variable_major else out.permute(1, 0, 2)
        h1 = ops.mnistcnn_sparse_fwd(x.view(-1, 784), P, sr, self.conv1.weight, self.conv1.bias, self.conv2.weight,
                                     self.conv2.bias, self.fc1.weight, self.fc1.bias, pre_gated=True,
                                     prep=None if torch.is_grad_enabled() else self._held_prep)
        out = ops.mlp(h1, [(self.fc2.weight, self.fc2.bias)], relu_in=True)       # gates h1's cotangent in its epilogue
        if variable_major:                       # [R, B, out_d]; a caller whose rows are already in the kernels' order
            out = out.view(sr.R, sr.B, -1)       # (sorted by crop origin) gets the output without a gather
            return out if sr.identity else ops.PermuteRowsFn.apply(out, sr.unsort, sr.order)
        return ops.PermuteRowsFn.apply(out.view(sr.R, sr.B, -1), sr.unsort, sr.order).permute(1, 0, 2)

    def supports_gated(self, x):
        """the gate of a DAG conditioner + this net's conv front as one autograd node (gnf_hip.ops.DagConvFrontFn): the
        28 x 28 net on a [B, 784] batch, one masked copy per pixel"""
        return x.dim() == 2 and self._fused_front(x)

    def forward_gated(self, x, A, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset):
        """embedding_net(e) for e[b*d+i] = x[b] * gate(importance(A[i])) (DAGConditioner.py:94-166,169), e built inside the
        fused front: [B*784, out_d]"""
        pooled = ops.dag_conv_front(x, A, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset,
                                    self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                    self.exact_pool_ties)
        return ops.mlp(pooled, _linears([self.fc1, self.fc2]))

    _held_prep = None

    def hold_prepared(self):
        """context manager for a caller that evaluates sparse_rows many times with unchanged parameters (the levels of one
        `invert`): the parameter-only tables of the sparse front are built once instead of once per call (two launches,
        ~28 us, of the ~60 us a level's front costs)"""
        import contextlib

        @contextlib.contextmanager
        def hold():
            prev = self._held_prep
            if prev is None and self.fc1.weight.is_cuda and self.fc1.weight.shape[1] == 2304 and self.fc1.weight.shape[0] % 4 == 0:
                self._held_prep = ops.mnistcnn_sparse_prepare(self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                                               self.fc1.weight, self.fc1.bias)
            try:
                yield
            finally:
                self._held_prep = prev
        return hold()

    def forward(self, x, context=None):
        rows = x.shape[0]
        if self._fused_front(x):
            feat = ops.MnistConvFn.apply(x.view(-1, 784), self.conv1.weight, self.conv1.bias, self.conv2.weight,
                                         self.conv2.bias, self.exact_pool_ties)
        elif self._embeddable(x):
            # the 14x14 / 7x7 scales of the multi-scale factory: the image sits in the top-left corner of a zero 28x28
            # one.  Valid convolutions never look across the corner's edge for the output positions that exist in the
            # small image, so conv1 / conv2 / pool agree there exactly and the fused kernels apply (4x / 16x the
            # necessary area, still several times cheaper than an im2col round trip through HBM)
            _, h, w = self.size_img
            big = F.pad(x.view(-1, 1, h, w), (0, 28 - w, 0, 28 - h)).view(-1, 784)
            pooled = ops.MnistConvFn.apply(big, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                           self.exact_pool_ties)
            feat = pooled.view(-1, 16, 12, 12)[:, :, :(h - 4) // 2, :(w - 4) // 2].reshape(rows, -1)
        else:
            # any other geometry: batched im2col + MFMA GEMM (no MIOpen: its find step costs minutes on this stack)
            img = x.view(-1, *self.size_img)
            feat = F.relu(_conv3x3(img, self.conv1.weight, self.conv1.bias))
            feat = F.max_pool2d(_conv3x3(feat, self.conv2.weight, self.conv2.bias), 2).flatten(1)
        return ops.mlp(feat, _linears([self.fc1, self.fc2])).view(rows, -1)


class CIFAR10CNN(nn.Module):
    """LeNet-style CIFAR embedding net of the reference (:51-72).  Not on any measured configuration: convolutions stay
    on torch, the fc chain on the MFMA GEMM."""

    def __init__(self, out_d=10, fc_l=[400, 128, 84], size_img=[3, 32, 32], k_size=5):
        super().__init__()
        self.size_img, self.out_d = size_img, out_d
        self.conv1 = nn.Conv2d(size_img[0], 6, k_size)
        self.pool = nn.MaxPool2d(2, 2)
        self.conv2 = nn.Conv2d(6, 16, k_size)
        self.fc1 = nn.Linear(fc_l[0], fc_l[1])
        self.fc2 = nn.Linear(fc_l[1], fc_l[2])
        self.fc3 = nn.Linear(fc_l[2], out_d)

    def forward(self, x, context=None):
        rows = x.shape[0]
        feat = x.view(-1, *self.size_img)
        for conv in (self.conv1, self.conv2):
            feat = self.pool(F.relu(conv(feat)))
        return ops.mlp(feat.reshape(rows, -1), _linears([self.fc1, self.fc2, self.fc3])).view(rows, -1)


class IdentityNN(nn.Module):
    def forward(self, x, context=None):
        return x
